import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lwpose_amd
from lwpose_amd import synth, workload
net, _ = workload.build_net(1, 1, 0)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(1, 368, 656))).cuda()
eng = net.engine
for _ in range(5):
    eng.infer_poses_async(x); r = eng.fetch_poses()
print("poses", len(r[0][0]), "kpts", len(r[0][1]))
