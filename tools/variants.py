"""kernel variant the launch plan picks for every layer: variants.py batch dtype [nref] (HIP box only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, lwpose_amd
from lwpose_amd import synth, workload
batch, dtype = int(sys.argv[1]), sys.argv[2]
nref = int(sys.argv[3]) if len(sys.argv) > 3 else 1
net, _ = workload.build_net(nref, 1, 0, dtype, 368, 656)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(batch, 368, 656))).cuda()
eng = net.engine
eng.infer_poses_async(x); eng.fetch_poses()
rows = {n: ms * 1e3 for n, kc, ms in eng.profile_launches(x, 10)}
for l in eng.layers():
    print("%-44s %-34s %8.1f us" % (l["name"], eng.layer_variant(l["index"]), rows.get(l["name"], float("nan"))))
