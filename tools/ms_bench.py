"""time of the multi-scale accumulate step (up-sample x8 + crop + resize + avg += m / n) at the config-4 geometries, fused kernel against
the two-kernel form: ms_bench.py [batch] (HIP box only)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time
sys.path.insert(0, %r)
import torch, lwpose_amd
from lwpose_amd.runtime import Engine
N = int(sys.argv[1])
eng = Engine(0)
# (h, w, pad) of the network maps for a 368 x 656 frame at scales 0.5 / 1 / 1.5 / 2 (val.py:84-93, stride 8)
geos = [("0.5", 46, 46, [92, 20, 92, 20]), ("1.0", 46, 82, [0, 0, 0, 0]), ("1.5", 69, 123, [0, 0, 0, 0]), ("2.0", 92, 164, [0, 0, 0, 0])]
tot = 0.0
for name, h, w, pad in geos:
    for C in (19, 38):
        maps = torch.rand(N, C, h, w, device="cuda") - 0.3
        acc = torch.empty(N, 368, 656, C, device="cuda")
        for _ in range(2): eng.multiscale_accumulate(acc, maps, 8, pad, 3, init=True)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): eng.multiscale_accumulate(acc, maps, 8, pad, 3, init=False)
        torch.cuda.synchronize(); us = (time.perf_counter() - t0) / 5 * 1e6
        tot += us if name != "2.0" else 0.0
        print("scale %%s C=%%2d: %%8.0f us  (%%.2f TB/s of accumulator traffic)" %% (name, C, us, 2 * acc.numel() * 4 / us / 1e6))
print("scales 0.5 + 1 + 1.5, both maps: %%.2f ms" %% (tot / 1e3))
''' % ROOT
batch = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].isdigit() else "32"
cfgs = [("fused, four channels per lane", {}), ("fused, one channel per lane (LWP_MS_VEC=0)", {"LWP_MS_VEC": "0"}), ("two kernels (LWP_MS_FUSED=0)", {"LWP_MS_FUSED": "0"})]
if "--tx" in sys.argv:
    cfgs = [("fused, %s-column tiles" % t, {"LWP_MS_TX": t}) for t in ("11", "13", "16", "26", "32")] + cfgs
for tag, env in cfgs:
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD, batch], capture_output=True, text=True, env=e, timeout=600)
    print("== " + tag); print(r.stdout if r.returncode == 0 else (r.stderr or r.stdout)[-800:])
