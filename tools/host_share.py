"""where the submitting thread spends its time in the headline loop (3 engine streams, batch 1): host_share.py [streams] (HIP box only)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, lwpose_amd
from lwpose_amd import synth, workload
from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
E = int(sys.argv[1]) if len(sys.argv) > 1 else 3
net, _ = workload.build_net(1, 1, 0, "fp32", 368, 656)
engines = [net.engine]
blob = torch.empty(net.engine.weights_blob_bytes(), dtype=torch.uint8, device="cuda")
net.engine.export_weights(blob)
keep = []
for _ in range(E - 1):
    n2 = PoseEstimationWithMobileNet(1); n2.eval().cuda(0); n2.engine.import_weights(blob); engines.append(n2.engine); keep.append(n2)
x = torch.from_numpy(workload.normalized_input(synth.make_frames(1, 368, 656))).cuda()
def run(k, stamps=None):
    pending = []
    for i in range(k):
        e, slot = engines[i % E], (i // E) & 1
        if len(pending) >= 2 * E:
            pe, ps = pending.pop(0)
            t0 = time.perf_counter(); pe.pipeline_fetch(ps); t1 = time.perf_counter()
            if stamps is not None: stamps["fetch"].append(t1 - t0)
        t0 = time.perf_counter(); e.pipeline_submit(x, slot, 4, True); t1 = time.perf_counter()
        if stamps is not None: stamps["submit"].append(t1 - t0)
        pending.append((e, slot))
    for pe, ps in pending: pe.pipeline_fetch(ps)
run(600)
st = {"fetch": [], "submit": []}
torch.cuda.synchronize(); t0 = time.perf_counter(); run(3000, st); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("streams %d: %.0f frames/s, %.1f us per frame; submit median %.1f us (mean %.1f), fetch median %.1f us (mean %.1f)" % (
    E, 3000 / dt, dt / 3000 * 1e6, np.median(st["submit"]) * 1e6, np.mean(st["submit"]) * 1e6, np.median(st["fetch"]) * 1e6, np.mean(st["fetch"]) * 1e6))
