"""CPU: pins oracle/ (the CPU restatement) to outputs captured from the reference itself
(tests/golden/*, written by oracle/make_golden.py).  Bit-exact for the integer / float64
post-processing; <= 1e-5 max-abs for the fp32 network (same torch CPU kernels)."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

import lwpose_amd  # noqa: F401
from lwpose_amd import arch, synth
from oracle import net_ref, post_ref

from conftest import GOLDEN


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def net_input(n, h, w, seed):
    fr = synth.make_frames(n, h, w, seed0=seed)
    x = (fr.astype(np.float32) - 128.0) * np.float32(1 / 256)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2))


@pytest.mark.parametrize("nref", [1, 3])
def test_param_table_matches_reference_keys(nref):
    keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys_nref%d.json" % nref)))
    tab = arch.param_table(nref)
    assert [k for k, _, _ in keys] == [p.key for p in tab]
    assert [tuple(s) for _, s, _ in keys] == [tuple(p.shape) for p in tab]
    sd = synth.make_state_dict(nref)
    for k, s, dt in keys:
        assert tuple(sd[k].shape) == tuple(s) and str(sd[k].dtype) == dt, k


@pytest.mark.parametrize("nref", [1, 3])
def test_net_small_all_stages_and_taps(nref):
    g = np.load(os.path.join(GOLDEN, "net_small_nref%d.npz" % nref))
    x = net_input(2, 64, 96, seed=100)
    assert digest(x) == str(g["in_digest"])
    taps = {}
    outs = net_ref.forward(synth.make_state_dict(nref, seed=1), torch.from_numpy(x), nref, taps)
    assert len(outs) == 2 * (1 + nref)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g["out%d" % i], rtol=0, atol=2e-5)
    checked = 0
    for k in g.files:
        if k.startswith("tap:") and k[4:] in taps:
            np.testing.assert_allclose(taps[k[4:]].numpy().reshape(-1)[::7], g[k], rtol=0,
                                       atol=2e-5 * max(1.0, float(np.abs(g[k]).max())), err_msg=k)
            checked += 1
    assert checked >= 15 + 5 * nref


def test_net_mid_batch3():
    g = np.load(os.path.join(GOLDEN, "net_mid_nref1.npz"))
    x = net_input(3, 184, 328, seed=200)
    assert digest(x) == str(g["in_digest"])
    outs = net_ref.forward(synth.make_state_dict(1, seed=7), torch.from_numpy(x), 1)
    for i, o in enumerate(outs):
        np.testing.assert_allclose(o.numpy(), g["out%d" % i], rtol=0, atol=2e-5)


def test_net_full_368x656():
    g = np.load(os.path.join(GOLDEN, "net_full_nref1.npz"))
    x = net_input(1, 368, 656, seed=0)
    assert digest(x) == str(g["in_digest"])
    outs = net_ref.forward(synth.make_state_dict(1, seed=1, head_gain=4.0), torch.from_numpy(x), 1)
    for i, o in enumerate(outs):
        o = o.numpy()
        assert tuple(o.shape) == tuple(g["out%d_shape" % i])
        np.testing.assert_allclose(o.reshape(-1)[::5], g["out%d_sample" % i], rtol=0, atol=2e-5)
        assert abs(o.astype(np.float64).sum() - float(g["out%d_sum" % i])) < 1e-2


def test_extract_adversarial_bit_exact():
    g = np.load(os.path.join(GOLDEN, "extract_adversarial.npz"))
    names = [k[3:] for k in g.files if k.startswith("in:")]
    assert len(names) >= 8
    for nm in names:
        hm = g["in:" + nm].copy()
        lst = []
        n = post_ref.extract_keypoints(hm, lst, 17)
        assert n == int(g["n:" + nm]), nm
        got = np.array([[p[0], p[1], p[2], p[3]] for p in lst[0]], dtype=np.float64).reshape(-1, 4)
        assert np.array_equal(got, g["kp:" + nm]), nm
        assert np.array_equal(hm, g["mut:" + nm], equal_nan=True), nm


def _by_type_from(kp):
    bt = [[] for _ in range(18)]
    for x, y, s, i, t in kp:
        bt[int(t)].append((np.int64(x), np.int64(y), np.float32(s), int(i)))
    return bt


def test_group_adversarial_bit_exact():
    g = np.load(os.path.join(GOLDEN, "group_adversarial.npz"))
    names = [k[4:] for k in g.files if k.startswith("paf:")]
    assert len(names) >= 7
    for nm in names:
        for tag, demo in (("demo", True), ("val", False)):
            ent, allk = post_ref.group_keypoints(_by_type_from(g["kp:" + nm]), g["paf:" + nm], demo=demo)
            key = "%s:%s" % (nm, tag)
            assert tuple(np.asarray(ent).shape) == tuple(g["ent_shape:" + key]), key
            assert tuple(np.asarray(allk).shape) == tuple(g["allk_shape:" + key]), key
            assert np.array_equal(np.asarray(ent, dtype=np.float64), g["ent:" + key]), key
            assert np.array_equal(np.asarray(allk, dtype=np.float64), g["allk:" + key]), key


def run_oracle_post(heat_up, paf_up, demo):
    heat = heat_up.copy()
    by_type, total = [], 0
    for k in range(18):
        total += post_ref.extract_keypoints(heat[:, :, k], by_type, total)
    ent, allk = post_ref.group_keypoints(by_type, paf_up, demo=demo)
    kp = np.array([[p[0], p[1], p[2], p[3], t] for t, l in enumerate(by_type) for p in l], dtype=np.float64)
    return heat, kp.reshape(-1, 5), np.asarray(ent, dtype=np.float64), np.asarray(allk, dtype=np.float64)


POST = ["p1_small", "p3_small", "p0_empty", "p5_mid", "p10_full", "p4_noisy", "p2_r8"]


@pytest.mark.parametrize("name", POST)
def test_post_pipeline_bit_exact(name):
    g = np.load(os.path.join(GOLDEN, "post_%s.npz" % name))
    n, h, w, seed, ratio = [int(v) for v in g["params"]]
    heat, paf, _ = synth.make_pose_maps(n, h, w, seed, float(g["drop"]), float(g["noise"]))
    assert digest(heat) + digest(paf) == str(g["lowres_digest"])
    hu = post_ref.upsample_cubic(heat.transpose(1, 2, 0), ratio)
    pu = post_ref.upsample_cubic(paf.transpose(1, 2, 0), ratio)
    assert digest(hu) + digest(pu) == str(g["up_digest"])
    for tag, demo in (("demo", True), ("val", False)):
        hm, kp, ent, allk = run_oracle_post(hu, pu, demo)
        assert np.array_equal(kp, g[tag + "_kp"])
        assert tuple(ent.shape) == tuple(g[tag + "_entries_shape"])
        assert np.array_equal(ent, g[tag + "_entries"])
        assert np.array_equal(allk, g[tag + "_allk"])
    assert digest(hm[:, :, :18]) == str(g["mutated_digest"])
    if n >= 3:   # the synthetic people are actually found (the fixture is not vacuous)
        assert g["demo_entries"].shape[0] >= n - 1


def test_linspace_truncation_table():
    g = np.load(os.path.join(GOLDEN, "linspace2d.npz"))
    starts, deltas = g["starts"], g["deltas"]
    step = (1 / 9) * deltas.astype(np.int64)
    x = step[None, :, None] * np.arange(10) + starts[:, None, None]
    assert np.array_equal(np.trunc(x).astype(np.int64), g["trunc"].astype(np.int64))
    assert np.array_equal(np.round(x).astype(np.int64), g["round"].astype(np.int64))


def test_upsample_matches_torch_bicubic():
    """cv2 is absent: cross-check the restated OpenCV float bicubic against torch's (same A=-0.75 kernel)."""
    heat, _, _ = synth.make_pose_maps(3, 24, 40, 11)
    for ratio in (4, 8):
        up = post_ref.upsample_cubic(heat.transpose(1, 2, 0), ratio)
        t = torch.nn.functional.interpolate(torch.from_numpy(heat)[None], scale_factor=ratio, mode="bicubic",
                                            align_corners=False)[0].numpy().transpose(1, 2, 0)
        assert up.shape == t.shape
        assert np.abs(up - t).max() < 2e-6
    # the x4 / x8 phases are dyadic, so OpenCV's A = -0.75 kernel is exact in float32 at every one of them
    assert np.array_equal(post_ref.cubic_coeffs(0.625), np.array([-0.06591796875, 0.42626953125, 0.74951171875, -0.10986328125], np.float32))

    def kernel(t, a=-0.75):
        t = abs(t)
        return (a + 2) * t ** 3 - (a + 3) * t ** 2 + 1 if t <= 1 else a * t ** 3 - 5 * a * t ** 2 + 8 * a * t - 4 * a
    for fr in (0.125, 0.375, 0.625, 0.875, 0.0625, 0.5625):
        c = post_ref.cubic_coeffs(fr)
        assert np.array_equal(c, np.array([kernel(fr + 1), kernel(fr), kernel(1 - fr), kernel(2 - fr)], np.float32))
        assert float(c.astype(np.float64).sum()) == 1.0


def test_generic_resize_matches_integer_ratio_and_torch():
    heat, _, _ = synth.make_pose_maps(3, 23, 41, 11)
    hwc = heat.transpose(1, 2, 0)
    assert np.array_equal(post_ref.resize_cubic(hwc, 41 * 4, 23 * 4), post_ref.upsample_cubic(hwc, 4))
    for (dw, dh) in ((100, 57), (30, 17), (41, 23)):
        got = post_ref.resize_cubic(hwc, dw, dh)
        t = torch.nn.functional.interpolate(torch.from_numpy(heat)[None], size=(dh, dw), mode="bicubic",
                                            align_corners=False)[0].numpy().transpose(1, 2, 0)
        assert np.abs(got - t).max() < 3e-6
    assert np.array_equal(post_ref.resize_cubic(hwc, 41, 23), hwc)          # identity size -> identity
