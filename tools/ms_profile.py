"""Time split of one BASELINE-config-4 step (batch 32, nref 3, scales 0.5/1.0/1.5): run under rocprofv3 --kernel-trace --stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lwpose_amd
from lwpose_amd import synth, workload, val as lwval
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
net3, _ = workload.build_net(3, 1, 0, "fp32", 368, 656)
fr = list(synth.make_frames(8, 368, 656))
ins = [(torch.from_numpy(np.tile(xs, (B // 8, 1, 1, 1))).cuda(), pad) for xs, pad in lwval.scaled_inputs(fr, [0.5, 1.0, 1.5], 368, 8)]
def step():
    t0 = time.perf_counter()
    ah, ap = lwval.accumulate_scales(net3, ins, 368, 656, 8)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    r = lwval.poses_batch(net3, ah, ap)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return t1 - t0, t2 - t1
step()
for _ in range(2):
    a, b = step()
    print("accumulate_scales %.1f ms, poses_batch %.1f ms" % (a * 1e3, b * 1e3))
eng = net3.engine
for x, pad in ins:
    eng.infer_poses(x[:1])  # warm
    ms = eng.time_pipeline(x, 3, what=0) / 3
    print("forward only", tuple(x.shape), "%.1f ms" % ms)
# per-call split (find which call is slow on a slow box)
dev = torch.device("cuda", 0)
for rep in range(2):
    ah = torch.empty((B, 368, 656, 19), dtype=torch.float32, device=dev); ap = torch.empty((B, 368, 656, 38), dtype=torch.float32, device=dev)
    for k, (x, pad) in enumerate(ins):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        outs = net3(x); eng.synchronize(); t1 = time.perf_counter()
        eng.multiscale_accumulate(ah, outs[-2], 8, pad, 3, init=(k == 0)); t2 = time.perf_counter()
        eng.multiscale_accumulate(ap, outs[-1], 8, pad, 3, init=(k == 0)); t3 = time.perf_counter()
        print("rep %d scale %d: net %.1f ms, acc heat %.1f ms, acc paf %.1f ms" % (rep, k, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
