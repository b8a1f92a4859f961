"""Deterministic synthetic weights, frames and pose maps (no torch RNG streams).

There is no trained checkpoint offline (reference README.md:95 is a download), so
every test/bench input is produced by a counter-based generator that can be
re-implemented anywhere: splitmix64 over (stream seed, element index).
"""
from collections import OrderedDict

import numpy as np

from .arch import param_table

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
_HEAD_BASE_GAIN = 0.1       # final 1x1 of each heat/PAF head: maps land in the trained range (|v| ~ 0.1..1)
_RESIDUAL_IN_GAIN = 0.55    # refinement block entry 1x1


def splitmix64(index, seed):
    """splitmix64 output for state = seed + (index+1)*golden; vectorised over uint64 ``index``."""
    with np.errstate(over="ignore"):
        z = (np.uint64(seed) + (index.astype(np.uint64) + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def fnv1a64(text, seed=0):
    h = (0xCBF29CE484222325 ^ (seed * 0x9E3779B97F4A7C15)) & 0xFFFFFFFFFFFFFFFF
    for b in text.encode("utf-8"):
        h = ((h ^ b) * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF
    return h


def uniform(shape, stream_seed, lo=0.0, hi=1.0):
    """float32 uniform in [lo, hi): 24 random mantissa bits per element."""
    n = int(np.prod(shape)) if len(shape) else 1
    bits = splitmix64(np.arange(n, dtype=np.uint64), stream_seed) >> np.uint64(40)
    u = bits.astype(np.float64) * (1.0 / 16777216.0)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def make_state_dict(num_refinement_stages=1, seed=1, num_channels=128, num_heatmaps=19, num_pafs=38,
                    head_gain=1.0, heat_bias=0.0, as_torch=True):
    """Random-init weights in the reference's ``state_dict`` layout (keys, shapes, dtypes).

    conv weights ~ U(-a, a), a = sqrt(6 / fan_in) (keeps activations O(1) through ReLU stacks);
    BN gamma ~ U(0.5, 1.5), beta/mean ~ U(-0.1, 0.1), var ~ U(0.5, 1.5).
    The entry 1x1 of every refinement block and the last 1x1 of every head carry fixed extra gains
    (_RESIDUAL_IN_GAIN, _HEAD_BASE_GAIN) so that the stage outputs stay in the range of a trained net.
    ``head_gain`` scales the last 1x1 of every heat/PAF head and ``heat_bias`` is added to the
    heat-map head bias, so that maps cross the 0.1 peak threshold (reference keypoints.py:17).
    """
    sd = OrderedDict()
    for p in param_table(num_refinement_stages, num_channels, num_heatmaps, num_pafs):
        s = fnv1a64(p.key, seed)
        if p.role == "conv_w":
            a = float(np.sqrt(6.0 / p.fan_in))
            v = uniform(p.shape, s, -a, a)
            if p.key.endswith("heatmaps.1.0.weight") or p.key.endswith("pafs.1.0.weight"):
                v = (v * np.float32(head_gain * _HEAD_BASE_GAIN)).astype(np.float32)
            elif p.key.endswith(".initial.0.weight"):       # keeps the 5-block residual stack O(1)
                v = (v * np.float32(_RESIDUAL_IN_GAIN)).astype(np.float32)
        elif p.role == "conv_b":
            v = uniform(p.shape, s, -0.1, 0.1)
            if p.key.endswith("heatmaps.1.0.bias"):
                v = (v + np.float32(heat_bias)).astype(np.float32)
        elif p.role in ("bn_w", "bn_var"):
            v = uniform(p.shape, s, 0.5, 1.5)
        elif p.role in ("bn_b", "bn_mean"):
            v = uniform(p.shape, s, -0.1, 0.1)
        else:  # bn_nbt
            v = np.array(1, dtype=np.int64)
        sd[p.key] = v
    if as_torch:
        import torch
        sd = OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v)).reshape(v.shape)) for k, v in sd.items())
    return sd


def make_frames(batch, height=368, width=656, seed0=0, smooth=True):
    """(batch, height, width, 3) uint8 BGR frames; frame i uses seed ``seed0 + i``.

    ``smooth``: low-frequency content (bilinear-upsampled 1/16-resolution noise) plus fine
    noise, closer to camera frames than white noise; still fully deterministic."""
    out = np.empty((batch, height, width, 3), dtype=np.uint8)
    for i in range(batch):
        s = fnv1a64("frame", seed0 + i)
        if not smooth:
            out[i] = (uniform((height, width, 3), s) * 256.0).astype(np.uint8)
            continue
        gh, gw = height // 16 + 2, width // 16 + 2
        coarse = uniform((gh, gw, 3), s).astype(np.float64)
        ys = (np.arange(height) + 0.5) / 16.0
        xs = (np.arange(width) + 0.5) / 16.0
        y0 = np.floor(ys).astype(int); fy = (ys - y0)[:, None, None]
        x0 = np.floor(xs).astype(int); fx = (xs - x0)[None, :, None]
        c = (coarse[y0][:, x0] * (1 - fy) * (1 - fx) + coarse[y0][:, x0 + 1] * (1 - fy) * fx +
             coarse[y0 + 1][:, x0] * fy * (1 - fx) + coarse[y0 + 1][:, x0 + 1] * fy * fx)
        fine = uniform((height, width, 3), s ^ 0x5555).astype(np.float64)
        out[i] = np.clip((0.8 * c + 0.2 * fine) * 256.0, 0, 255).astype(np.uint8)
    return out


# skeleton tables follow the reference's kpt order (modules/keypoints.py:5-8); restated here for map synthesis
LIMB_KPTS = [(1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (1, 8), (8, 9), (9, 10), (1, 11),
             (11, 12), (12, 13), (1, 0), (0, 14), (14, 16), (0, 15), (15, 17), (2, 16), (5, 17)]
LIMB_PAFS = [(12, 13), (20, 21), (14, 15), (16, 17), (22, 23), (24, 25), (0, 1), (2, 3), (4, 5),
             (6, 7), (8, 9), (10, 11), (28, 29), (30, 31), (34, 35), (32, 33), (36, 37), (18, 19), (26, 27)]

# a canonical standing pose in unit coordinates (x right, y down), neck at origin
_CANON = np.array([
    [0.00, -0.30], [0.00, 0.00], [-0.25, 0.02], [-0.32, 0.40], [-0.35, 0.75], [0.25, 0.02],
    [0.32, 0.40], [0.35, 0.75], [-0.15, 0.85], [-0.17, 1.45], [-0.18, 2.00], [0.15, 0.85],
    [0.17, 1.45], [0.18, 2.00], [-0.06, -0.36], [0.06, -0.36], [-0.14, -0.32], [0.14, -0.32]])


def make_pose_maps(n_people, h=46, w=82, seed=0, drop_prob=0.1, noise=0.01):
    """Low-resolution (stride-8) heat-maps (19,h,w) and PAFs (38,h,w), float32, with ``n_people``
    synthetic skeletons: Gaussian key-point blobs and unit limb vector fields.  Returns
    (heat, paf, people) where people[i] is an (18,2) array of low-res (x,y) or NaN if dropped."""
    s = fnv1a64("posemaps", seed)
    r = uniform((n_people, 64), s).astype(np.float64)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    heat = np.zeros((19, h, w), np.float64)
    paf = np.zeros((38, h, w), np.float64)
    cnt = np.zeros((38, h, w), np.float64)
    people = []
    for i in range(n_people):
        scale = (0.10 + 0.12 * r[i, 0]) * h
        cx = (0.08 + 0.84 * r[i, 1]) * w
        cy = (0.15 + 0.25 * r[i, 2]) * h
        lean = (r[i, 3] - 0.5) * 0.5
        pts = _CANON * scale
        pts = np.stack([pts[:, 0] + lean * pts[:, 1] + cx, pts[:, 1] + cy], 1)
        pts += (r[i, 4:40].reshape(18, 2) - 0.5) * 0.15 * scale
        keep = r[i, 40:58] > drop_prob
        keep &= (pts[:, 0] > 1) & (pts[:, 0] < w - 2) & (pts[:, 1] > 1) & (pts[:, 1] < h - 2)
        pts[~keep] = np.nan
        people.append(pts)
        for k in range(18):
            if keep[k]:
                g = np.exp(-((xx - pts[k, 0]) ** 2 + (yy - pts[k, 1]) ** 2) / (2 * 0.9 ** 2))
                heat[k] = np.maximum(heat[k], g)
        for l, ((a, b), (c0, c1)) in enumerate(zip(LIMB_KPTS, LIMB_PAFS)):
            if not (keep[a] and keep[b]):
                continue
            v = pts[b] - pts[a]
            n = np.hypot(*v)
            if n < 1e-6:
                continue
            u = v / n
            t = (xx - pts[a, 0]) * u[0] + (yy - pts[a, 1]) * u[1]
            d = np.abs((xx - pts[a, 0]) * u[1] - (yy - pts[a, 1]) * u[0])
            m = (t >= -0.5) & (t <= n + 0.5) & (d <= 1.0)
            paf[c0][m] += u[0]; paf[c1][m] += u[1]
            cnt[c0][m] += 1; cnt[c1][m] += 1
    paf = np.where(cnt > 0, paf / np.maximum(cnt, 1), 0.0)
    heat[18] = 1.0 - heat[:18].max(0)
    nz = uniform((57, h, w), s ^ 0xABCDEF).astype(np.float64) - 0.5
    heat = heat + noise * nz[:19]
    paf = paf + noise * nz[19:]
    return heat.astype(np.float32), paf.astype(np.float32), people


def calibrate_heads(sd, heat_raw, paf_raw, num_refinement_stages=1, peaks_per_channel=10, score_scale=0.25,
                    paf_mean=0.3, paf_std=0.1):
    """Affine re-parameterisation of the LAST stage's final 1x1 convs so that random-init maps look like a
    trained net's to the post-processing: per heat-map channel about ``peaks_per_channel`` low-res pixels
    exceed the 0.1 threshold (reference keypoints.py:17); every PAF channel gets mean ``paf_mean`` and std
    ``paf_std`` so a realistic share of candidate limbs passes the line integral (keypoints.py:129-137).

    heat_raw (NH,h,w) / paf_raw (NP,h,w): final-stage outputs obtained with ``sd`` on a sample frame (from any
    forward provider).  Returns a new state dict; only 4 tensors change."""
    import torch
    p = "refinement_stages.%d" % (num_refinement_stages - 1) if num_refinement_stages > 0 else "initial_stage"
    out = OrderedDict(sd)
    heat_raw = np.asarray(heat_raw, dtype=np.float64)
    paf_raw = np.asarray(paf_raw, dtype=np.float64)

    def to_np(t):
        return t.detach().cpu().numpy().astype(np.float64) if hasattr(t, "detach") else np.asarray(t, dtype=np.float64)

    def put(key, arr, like):
        arr = np.ascontiguousarray(arr.astype(np.float32))
        out[key] = torch.from_numpy(arr) if hasattr(like, "detach") else arr

    w, b = to_np(sd[p + ".heatmaps.1.0.weight"]), to_np(sd[p + ".heatmaps.1.0.bias"])
    for c in range(heat_raw.shape[0]):
        v = heat_raw[c].ravel()
        q = np.quantile(v, max(0.0, 1.0 - peaks_per_channel / float(v.size)))
        g = score_scale / max(v.std(), 1e-12)
        w[c] *= g
        b[c] = (b[c] - q) * g + 0.1
    put(p + ".heatmaps.1.0.weight", w, sd[p + ".heatmaps.1.0.weight"])
    put(p + ".heatmaps.1.0.bias", b, sd[p + ".heatmaps.1.0.bias"])
    w, b = to_np(sd[p + ".pafs.1.0.weight"]), to_np(sd[p + ".pafs.1.0.bias"])
    for c in range(paf_raw.shape[0]):
        v = paf_raw[c].ravel()
        g = paf_std / max(v.std(), 1e-12)
        w[c] *= g
        b[c] = (b[c] - v.mean()) * g + paf_mean
    put(p + ".pafs.1.0.weight", w, sd[p + ".pafs.1.0.weight"])
    put(p + ".pafs.1.0.bias", b, sd[p + ".pafs.1.0.bias"])
    return out
