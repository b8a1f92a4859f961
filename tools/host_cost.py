"""Host-side cost of pipeline_submit / pipeline_fetch (is the streaming loop CPU-bound?)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import lwpose_amd
from lwpose_amd import synth, workload
from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
E = int(sys.argv[1]) if len(sys.argv) > 1 else 2
net, sd = workload.build_net(1, 1, 0)
engs = [net.engine]
blob = torch.empty(net.engine.weights_blob_bytes(), dtype=torch.uint8, device="cuda")
net.engine.export_weights(blob)
keep = []
for _ in range(E - 1):
    n2 = PoseEstimationWithMobileNet(1); n2.eval().cuda(); n2.engine.import_weights(blob); keep.append(n2); engs.append(n2.engine)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(1, 368, 656))).cuda()
ts = [0.0, 0.0]
def run(k):
    pend = []
    for i in range(k):
        e = engs[i % E]; slot = (i // E) & 1
        if len(pend) >= 2 * E:
            pe, ps = pend.pop(0); t = time.perf_counter(); pe.pipeline_fetch(ps); ts[1] += time.perf_counter() - t
        t = time.perf_counter(); e.pipeline_submit(x, slot); ts[0] += time.perf_counter() - t; pend.append((e, slot))
    for pe, ps in pend: pe.pipeline_fetch(ps)
run(30); torch.cuda.synchronize()
ts[:] = [0.0, 0.0]
t0 = time.perf_counter(); run(600); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("engines", E, "frames/s", 600 / dt, "submit us", ts[0] / 600 * 1e6, "fetch us (incl. wait)", ts[1] / 600 * 1e6)
# submit cost without back-pressure: one submit on an idle engine
torch.cuda.synchronize()
c = []
for i in range(20):
    t = time.perf_counter(); engs[0].pipeline_submit(x, 0); c.append(time.perf_counter() - t); engs[0].pipeline_fetch(0)
print("idle submit us", np.median(c) * 1e6)
# pure host cost of a fetch whose results are already complete
c = []
for i in range(50):
    engs[0].pipeline_submit(x, 0); torch.cuda.synchronize(); time.sleep(0.002)
    t = time.perf_counter(); engs[0].pipeline_fetch(0); c.append(time.perf_counter() - t)
print("completed-fetch us", np.median(c) * 1e6)
import cProfile, pstats
engs[0].pipeline_submit(x, 0); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable(); engs[0].pipeline_fetch(0); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
