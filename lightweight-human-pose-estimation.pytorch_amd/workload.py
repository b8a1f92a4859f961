"""Synthetic benchmark / smoke workload: seeded frames + random-init weights whose last head layer is
calibrated (synth.calibrate_heads) from the HIP network's own output statistics on frame 0."""
import numpy as np

from . import synth
from .models.with_mobilenet import PoseEstimationWithMobileNet
from .modules.load_state import load_state


def normalized_input(frames_u8):
    """(B,H,W,3) uint8 -> (B,3,H,W) float32 = (u8 - 128) / 256   (val.py:30-33 + demo.py:64)."""
    x = (frames_u8.astype(np.float32) - 128.0) * np.float32(1 / 256)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


def build_net(nref=1, seed=1, device=0, dtype="fp32", height=368, width=656, calibrate=True):
    """Returns (net on cuda:device, state_dict actually loaded)."""
    net = PoseEstimationWithMobileNet(num_refinement_stages=nref, dtype=dtype)
    sd = synth.make_state_dict(nref, seed=seed)
    load_state(net, {"state_dict": sd})
    net.eval().cuda(device)
    if calibrate:
        x0 = normalized_input(synth.make_frames(1, height, width, seed0=0))
        outs = net(x0)
        sd = synth.calibrate_heads(sd, outs[-2][0], outs[-1][0], nref)
        load_state(net, {"state_dict": sd})
        net.cuda(device)
    return net, sd
