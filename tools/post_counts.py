"""candidate counts of the grouping kernels on the bench workload: post_counts.py batch dtype (HIP box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lwpose_amd
from lwpose_amd import synth, workload
batch, dtype = int(sys.argv[1]), sys.argv[2]
net, _ = workload.build_net(1, 1, 0, dtype, 368, 656)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(batch, 368, 656))).cuda()
eng = net.engine
eng.infer_poses_async(x, 4, True); eng.fetch_poses()
pk, kp, cand, pick = [np.stack(v) for v in zip(*[eng.post_counts(f) for f in range(batch)])]
for nm, a in (("peaks / type", pk), ("key-points / type", kp), ("candidates / limb", cand), ("picked / limb", pick)):
    print("%-20s mean %6.1f  max %4d  > 64: %d of %d" % (nm, a.mean(), a.max(), int((a > 64).sum()), a.size))
