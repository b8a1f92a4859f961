import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lwpose_amd
from lwpose_amd import synth, workload
from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
from lwpose_amd.modules.load_state import load_state
net, sd = workload.build_net(1, 1, 0)
nets = [net]
for _ in range(int(sys.argv[1]) - 1 if len(sys.argv) > 1 else 1):
    n2 = PoseEstimationWithMobileNet(1); load_state(n2, {"state_dict": sd}); n2.eval().cuda(); nets.append(n2)
engs = [n.engine for n in nets]
x = torch.from_numpy(workload.normalized_input(synth.make_frames(1, 368, 656))).cuda()
def run(k):
    E = len(engs)
    pend = []
    for i in range(k):
        e = engs[i % E]; slot = (i // E) & 1
        if len(pend) >= 2 * E:
            pe, ps = pend.pop(0); pe.pipeline_fetch(ps)
        e.pipeline_submit(x, slot); pend.append((e, slot))
    for pe, ps in pend: r = pe.pipeline_fetch(ps)
    return r
run(20); torch.cuda.synchronize()
t0 = time.perf_counter(); r = run(400); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("engines", len(engs), "frames/s", 400 / dt, "ms/frame", dt / 400 * 1e3, "poses", len(r[0][0]))
