import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import lwpose_amd
from lwpose_amd import workload, synth
net, _ = workload.build_net(1, 1, 0)
eng = net.engine
for b in (2, 4, 8, 16):
    x = torch.from_numpy(workload.normalized_input(synth.make_frames(b, 368, 656))).cuda()
    eng.infer_poses(x); eng.infer_poses(x)
    print("batch", b, "device ms/step", round(eng.time_pipeline(x, 10, what=0) / 10, 3), "frames/s (net only)", round(b / (eng.time_pipeline(x, 10, what=0) / 10) * 1e3))
