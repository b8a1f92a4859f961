"""host-frame boundary probe: cost of preprocess_u8 (pinned staging + DMA + resize kernel) against the resident step (HIP box only)"""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lwpose_amd
from lwpose_amd import synth, workload
net, _ = workload.build_net(1, 1, 0, "fp32", 368, 656)
eng = net.engine
frame = synth.make_frames(1, 368, 656)[0]
x = torch.from_numpy(workload.normalized_input(frame[None])).cuda()
def t(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
def pre_only():
    eng.preprocess_u8(frame, 368, 8, hand_over=False)
def pre_call_only(n=300):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); eng.preprocess_u8(frame, 368, 8, hand_over=False); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
def full():
    xx, sc, pad = eng.preprocess_u8(frame, 368, 8, hand_over=False)
    eng.infer_poses_async(xx, 4, True); return eng.fetch_poses()
def resident():
    eng.infer_poses_async(x, 4, True); return eng.fetch_poses()
dst = np.empty_like(frame)
print("np copy 720KB us", t(lambda: np.copyto(dst, frame)))
print("preprocess_u8 to completion us", t(pre_only))
print("preprocess_u8 call (host time, gpu idle) us", pre_call_only())
print("resident step us", t(resident))
print("u8 step us", t(full))
xd = torch.from_numpy(frame).cuda()
print("preprocess_u8 device frame us", t(lambda: eng.preprocess_u8(xd, 368, 8, hand_over=False)))
