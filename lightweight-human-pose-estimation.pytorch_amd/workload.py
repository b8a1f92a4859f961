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


def build_net(nref=1, seed=1, device=0, dtype="fp32", height=368, width=656, calibrate=True, multiscale=None):
    """Returns (net on cuda:device, state_dict actually loaded).

    ``calibrate``: the last stage's final 1x1 convs are re-parameterised (synth.calibrate_heads) from the network's own
    output statistics on frame 0 so that the maps cross the 0.1 peak threshold like a trained net's.
    ``multiscale``: list of scales (BASELINE config 4: [0.5, 1.0, 1.5]) — the statistics are then taken from the AVERAGED
    full-resolution maps of val.infer (val.py:81-110), which is what that config's extract/group stage reads; the affine
    head transform commutes with the (partition-of-unity) cubic resizes and the average."""
    net = PoseEstimationWithMobileNet(num_refinement_stages=nref, dtype=dtype)
    sd = synth.make_state_dict(nref, seed=seed)
    load_state(net, {"state_dict": sd})
    net.eval().cuda(device)
    if calibrate:
        frame0 = synth.make_frames(1, height, width, seed0=0)
        if multiscale:
            from . import val
            ah, ap = val.infer_batch(net, frame0, list(multiscale), height, 8)
            heat = ah[0].permute(2, 0, 1).cpu().numpy()
            paf = ap[0].permute(2, 0, 1).cpu().numpy()
            sd = synth.calibrate_heads(sd, heat, paf, nref, peaks_per_channel=10 * 64)   # same fraction of a x8-finer grid
        else:
            outs = net(normalized_input(frame0))
            sd = synth.calibrate_heads(sd, outs[-2][0], outs[-1][0], nref)
        load_state(net, {"state_dict": sd})
        net.cuda(device)
    return net, sd
