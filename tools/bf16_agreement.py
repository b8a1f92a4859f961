"""Measures what the bf16 conv stack (BASELINE config 3) changes against the fp32 CPU oracle on the calibrated 368x656 workload:
per-tensor max / mean abs error of the stage outputs, and skeleton agreement (pose count per frame, key-points of the same
type within 1 px).  Prints one JSON object; tests/test_gpu_parity.py pins its figures x 1.5.

    python tools/bf16_agreement.py [frames]
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lwpose_amd  # noqa: E402,F401
from lwpose_amd import synth, workload  # noqa: E402
from oracle import net_ref, post_ref  # noqa: E402


def oracle_post(heat_chw, paf_chw):
    hu = post_ref.upsample_cubic(heat_chw.transpose(1, 2, 0), 4)
    pu = post_ref.upsample_cubic(paf_chw.transpose(1, 2, 0), 4)
    by_type, total = [], 0
    for k in range(18):
        total += post_ref.extract_keypoints(hu[:, :, k], by_type, total)
    ent, allk = post_ref.group_keypoints(by_type, pu, demo=True)
    counts = np.array([len(b) for b in by_type])
    return np.asarray(ent, dtype=np.float64).reshape(-1, 20), np.asarray(allk, dtype=np.float64).reshape(-1, 4), counts


def by_type_lists(allk, counts):
    out, r = [], 0
    for t in range(18):
        out.append(allk[r:r + counts[t], :2])
        r += counts[t]
    return out


def match_fraction(a_lists, b_lists, tol=1):
    """share of a's key-points that have a same-type key-point of b within `tol` px (Chebyshev)."""
    hit = tot = 0
    for a, b in zip(a_lists, b_lists):
        tot += len(a)
        if len(a) and len(b):
            d = np.abs(a[:, None, :] - b[None, :, :]).max(axis=2)
            hit += int((d.min(axis=1) <= tol).sum())
    return hit, tot


def measure(n_frames=4, seed0=300):
    net, sd = workload.build_net(nref=1, seed=1, device=0, dtype="bf16")
    fr = synth.make_frames(n_frames, 368, 656, seed0=seed0)
    x = workload.normalized_input(fr)
    outs = net(x)
    ref = net_ref.forward(sd, torch.from_numpy(x), 1)
    tens = {}
    for i, (o, r) in enumerate(zip(outs, ref)):
        r = r.numpy()
        tens["out%d" % i] = {"max_abs": float(np.abs(o - r).max()), "mean_abs": float(np.abs(o - r).mean()), "ref_max": float(np.abs(r).max())}
    res = net.engine.infer_poses(x, 4, demo=True)
    h1 = t1 = h2 = t2 = 0
    poses = []
    for f in range(n_frames):
        ent, allk, counts = oracle_post(ref[-2][f].numpy(), ref[-1][f].numpy())
        e, a, c = res[f]
        ga, gb = by_type_lists(allk, counts), by_type_lists(a, c)
        u, v = match_fraction(ga, gb); h1 += u; t1 += v
        u, v = match_fraction(gb, ga); h2 += u; t2 += v
        poses.append((len(ent), len(e)))
    return {"frames": n_frames, "tensors": tens, "oracle_kpts_matched_by_bf16": h1 / max(t1, 1), "bf16_kpts_matched_by_oracle": h2 / max(t2, 1),
            "oracle_kpts": t1, "bf16_kpts": t2, "poses_oracle_vs_bf16": poses}


if __name__ == "__main__":
    print(json.dumps(measure(int(sys.argv[1]) if len(sys.argv) > 1 else 4)))
