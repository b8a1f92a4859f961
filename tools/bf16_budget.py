"""bf16 error budget of the conv stack (BASELINE config 3), on the CPU: which rounding points cost how much.

The HIP bf16 path rounds to bf16 (a) every packed GEMM weight, (b) every activation tensor it stores or hands to an MFMA (layer
outputs, the depthwise result inside a fused block, the hidden tensor of a head pair, the heat / PAF copies in the concat
buffer); accumulation, bias, activation functions and residual adds are f32.  This tool EMULATES that on the CPU (torch f32 convs
on operands rounded to bf16: bf16 x bf16 products are exact in f32, so only the summation order differs from the MFMA) and
switches the rounding points off group by group, measuring the stage-output error against the fp32 oracle and the key-point /
pose agreement through the oracle's post-processing.  It needs no GPU; tests/ pin the emulation to the HIP kernels' figures.

    python tools/bf16_budget.py [frames] > profiles/<round>/bf16_budget.json
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import lwpose_amd  # noqa: E402,F401
from lwpose_amd import synth, workload  # noqa: E402
from oracle import net_ref, post_ref  # noqa: E402

_BACKBONE = net_ref._BACKBONE
GROUPS = ("weights", "stem_out", "backbone_dw", "backbone_pw", "cpm", "initial_trunk", "heads_hidden", "concat_maps",
          "refine_initial", "refine_trunk", "last_hidden", "last_weights")


def rb(t):
    return t.to(torch.bfloat16).to(torch.float32)


def fold_bn(sd, conv, bn, has_bias):
    w = sd[conv + ".weight"].double()
    b = sd[conv + ".bias"].double() if has_bias else torch.zeros(w.shape[0], dtype=torch.float64)
    if bn:
        sc = sd[bn + ".weight"].double() / torch.sqrt(sd[bn + ".running_var"].double() + 1e-5)
        w = w * sc.view(-1, 1, 1, 1)
        b = (b - sd[bn + ".running_mean"].double()) * sc + sd[bn + ".bias"].double()
    return w.float(), b.float()


def forward_emulated(sd, x, nref, on):
    """``on``: set of GROUPS whose rounding is ACTIVE.  Returns the stage outputs (f32, like the HIP path's NCHW outputs)."""
    def W(conv, bn=None, bias=False, group="weights"):
        w, b = fold_bn(sd, conv, bn, bias)
        return (rb(w) if group in on else w), b

    def A(t, group):
        return rb(t) if group in on else t

    with torch.no_grad():
        w, b = fold_bn(sd, "model.0.0", "model.0.1", False)          # the stem computes in f32 from the f32 frame
        t = A(F.relu(F.conv2d(x, w, b, 2, 1)), "stem_out")
        for i, (s, d) in enumerate(_BACKBONE, start=1):
            c = t.shape[1]
            w, b = fold_bn(sd, "model.%d.0" % i, "model.%d.1" % i, False)           # depthwise weights stay f32
            t = A(F.relu(F.conv2d(t, w, b, s, d, d, c)), "backbone_dw")
            w, b = W("model.%d.3" % i, "model.%d.4" % i)
            t = A(F.relu(F.conv2d(t, w, b)), "backbone_pw")
        w, b = W("cpm.align.0", None, True)
        a = A(F.relu(F.conv2d(t, w, b)), "cpm")
        t = a
        for j in range(3):
            w, b = fold_bn(sd, "cpm.trunk.%d.0" % j, None, False)
            t = A(F.elu(F.conv2d(t, w, None, 1, 1, 1, t.shape[1])), "cpm")
            w, b = W("cpm.trunk.%d.2" % j)
            t = F.elu(F.conv2d(t, w, None))
            if j == 2:
                t = t + a                                             # residual in f32, one rounding
            t = A(t, "cpm")
        w, b = W("cpm.conv.0", None, True)
        feat = A(F.relu(F.conv2d(t, w, b, 1, 1)), "cpm")
        t = feat
        for j in range(3):
            w, b = W("initial_stage.trunk.%d.0" % j, None, True)
            t = A(F.relu(F.conv2d(t, w, b, 1, 1)), "initial_trunk")

        def heads(t, p, last):
            outs = []
            for nm in ("heatmaps", "pafs"):
                w0, b0 = W(p + ".%s.0.0" % nm, None, True)
                hdn = A(F.relu(F.conv2d(t, w0, b0)), "last_hidden" if last else "heads_hidden")
                w1, b1 = W(p + ".%s.1.0" % nm, None, True, "last_weights" if last else "weights")
                outs.append(F.conv2d(hdn, w1, b1))
            return outs
        outs = heads(t, "initial_stage", nref == 0)
        for k in range(nref):
            p = "refinement_stages.%d" % k
            t = torch.cat([feat, A(outs[-2], "concat_maps"), A(outs[-1], "concat_maps")], 1)
            for bl in range(5):
                q = "%s.trunk.%d" % (p, bl)
                w, b = W(q + ".initial.0", None, True)
                ini = A(F.relu(F.conv2d(t, w, b)), "refine_initial")
                w, b = W(q + ".trunk.0.0", q + ".trunk.0.1", True)
                u = A(F.relu(F.conv2d(ini, w, b, 1, 1)), "refine_trunk")
                w, b = W(q + ".trunk.1.0", q + ".trunk.1.1", True)
                u = F.relu(F.conv2d(u, w, b, 1, 2, 2))
                t = A(ini + u, "refine_trunk")
            outs.extend(heads(t, p, k == nref - 1))
    return outs


def oracle_post(heat_chw, paf_chw):
    hu = post_ref.upsample_cubic(heat_chw.transpose(1, 2, 0), 4)
    pu = post_ref.upsample_cubic(paf_chw.transpose(1, 2, 0), 4)
    by_type, total = [], 0
    for k in range(18):
        total += post_ref.extract_keypoints(hu[:, :, k], by_type, total)
    ent, allk = post_ref.group_keypoints(by_type, pu, demo=True)
    return len(ent), [np.asarray(b, dtype=np.float64).reshape(-1, 4)[:, :2] if len(b) else np.zeros((0, 2)) for b in by_type]


def match_fraction(a_lists, b_lists, tol=1):
    hit = tot = 0
    for a, b in zip(a_lists, b_lists):
        tot += len(a)
        if len(a) and len(b):
            d = np.abs(a[:, None, :] - b[None, :, :]).max(axis=2)
            hit += int((d.min(axis=1) <= tol).sum())
    return hit, tot


def calibrated_state(nref=1, seed=1, height=368, width=656):
    """The benchmark's calibrated weights without a GPU: head statistics from the fp32 ORACLE on frame 0 (workload.build_net takes
    them from the HIP network, which agrees with the oracle to ~1e-5)."""
    sd = synth.make_state_dict(nref, seed=seed)
    x0 = torch.from_numpy(workload.normalized_input(synth.make_frames(1, height, width, seed0=0)))
    outs = net_ref.forward(sd, x0, nref)
    return synth.calibrate_heads(sd, outs[-2][0].numpy(), outs[-1][0].numpy(), nref)


def evaluate(sd, x, nref, on, ref, ref_post):
    outs = forward_emulated(sd, x, nref, on)
    tens = {}
    for i, (o, r) in enumerate(zip(outs, ref)):
        sc = max(1.0, float(r.abs().max()))
        tens["out%d" % i] = {"max_abs_over_scale": float((o - r).abs().max()) / sc, "mean_abs_over_scale": float((o - r).abs().mean()) / sc}
    h1 = t1 = h2 = t2 = 0
    poses = []
    for f in range(x.shape[0]):
        n, lists = oracle_post(outs[-2][f].numpy(), outs[-1][f].numpy())
        n0, lists0 = ref_post[f]
        u, v = match_fraction(lists0, lists); h1 += u; t1 += v
        u, v = match_fraction(lists, lists0); h2 += u; t2 += v
        poses.append((n0, n))
    return {"tensors": tens, "oracle_kpts_matched": h1 / max(t1, 1), "emulated_kpts_matched": h2 / max(t2, 1), "poses_oracle_vs_emulated": poses}


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    nref = 1
    sd = calibrated_state(nref)
    sd = {k: (v if hasattr(v, "detach") else torch.from_numpy(np.asarray(v))) for k, v in sd.items()}
    x = torch.from_numpy(workload.normalized_input(synth.make_frames(n_frames, 368, 656, seed0=300)))
    ref = net_ref.forward(sd, x, nref)
    ref_post = [oracle_post(ref[-2][f].numpy(), ref[-1][f].numpy()) for f in range(n_frames)]
    allg = set(GROUPS)
    rows = {"all_bf16 (= the HIP bf16 path)": allg, "nothing rounded (sanity: the oracle itself)": set()}
    for g in GROUPS:
        rows["only " + g] = {g}
    for g in GROUPS:
        rows["all but " + g] = allg - {g}
    rows["all but weights + last_weights"] = allg - {"weights", "last_weights"}
    rows["all but last_hidden + last_weights (final head pair in f32)"] = allg - {"last_hidden", "last_weights"}
    rows["all but last_hidden + last_weights + concat_maps + refine_initial"] = allg - {"last_hidden", "last_weights", "concat_maps", "refine_initial"}
    rows["activations only (all weights f32)"] = allg - {"weights", "last_weights"}
    rows["weights only (all activations f32)"] = {"weights", "last_weights"}
    out = {"frames": n_frames, "workload": "calibrated 368x656, nref 1, frames seed 300.. (as tools/bf16_agreement.py)", "rows": {}}
    for name, on in rows.items():
        out["rows"][name] = evaluate(sd, x, nref, on, ref, ref_post)
        r = out["rows"][name]
        print("%-75s max %.4f %.4f %.4f %.4f  mean(out2) %.5f  kpts %.3f / %.3f  poses %s" % (
            name, *[r["tensors"]["out%d" % i]["max_abs_over_scale"] for i in range(4)], r["tensors"]["out2"]["mean_abs_over_scale"],
            r["oracle_kpts_matched"], r["emulated_kpts_matched"], r["poses_oracle_vs_emulated"]), file=sys.stderr, flush=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
