"""BASELINE config 4 steps for profiling: batch 32, 3 refinement stages, scales [0.5, 1.0, 1.5], 32 uint8 frames resident in HBM ->
poses on the host (val.py:81-134), exactly what bench.py's other_configs.batch32_nref3_multiscale_fp32 times.
    rocprofv3 --kernel-trace --stats -- python3 tools/cfg4_step.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lwpose_amd
from lwpose_amd import synth, workload, val as lwval
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
scales = [0.5, 1.0, 1.5]
net3, _ = workload.build_net(3, 1, 0, "fp32", 368, 656, multiscale=scales)
fr = torch.from_numpy(synth.make_frames(32, 368, 656, seed0=500)).cuda()
def step():
    ah, ap_ = lwval.infer_batch(net3, fr, scales, 368, 8)
    return lwval.poses_batch(net3, ah, ap_)
step()
for i in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = step()
    torch.cuda.synchronize()
    print("step %d: %.1f ms, %.1f poses / frame" % (i, (time.perf_counter() - t0) * 1e3, float(np.mean([len(x[0]) for x in r]))))
