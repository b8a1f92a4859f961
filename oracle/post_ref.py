"""ORACLE — test infrastructure, NOT product code.

NumPy restatement of the reference's post-processing: bicubic up-sampling of the stage
outputs, key-point extraction (threshold + strict 4-neighbour maximum + greedy radius-6
suppression), PAF line-integral pair scoring, greedy matching and pose assembly.
Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg import it.

Follows (reference file:line):
  demo.py:70-76                ×4 ``cv2.resize(..., INTER_CUBIC)`` of HWC float maps
  modules/keypoints.py:5-8     skeleton tables
  modules/keypoints.py:11-13   linspace2d (float64, multiply then add)
  modules/keypoints.py:16-48   extract_keypoints
  modules/keypoints.py:51-201  group_keypoints

Pinning: extract/group are checked against outputs captured from the reference itself
(oracle/make_golden.py -> tests/golden/post_*.npz).  The bicubic resize lives in OpenCV
(opencv-python>=3.4.0.14, requirements.txt:4), which is absent here: it is restated from
OpenCV's published float algorithm (Keys kernel A=-0.75, src = (dst+0.5)/r-0.5, replicated
border, horizontal pass then vertical pass, left-to-right float32 sums) and cross-checked
against torch's bicubic (same kernel) to 1e-6 — "parity unpinned vs cv2".
"""
import numpy as np

KPT_IDS = [[1, 2], [1, 5], [2, 3], [3, 4], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11],
           [11, 12], [12, 13], [1, 0], [0, 14], [14, 16], [0, 15], [15, 17], [2, 16], [5, 17]]
PAF_IDS = [[12, 13], [20, 21], [14, 15], [16, 17], [22, 23], [24, 25], [0, 1], [2, 3], [4, 5],
           [6, 7], [8, 9], [10, 11], [28, 29], [30, 31], [34, 35], [32, 33], [36, 37], [18, 19], [26, 27]]


# ----------------------------------------------------------------------------- bicubic
def cubic_coeffs(frac):
    """OpenCV ``interpolateCubic`` in float32 (A = -0.75)."""
    f = np.float32
    x = f(frac)
    A = f(-0.75)
    c0 = ((A * (x + f(1)) - f(5) * A) * (x + f(1)) + f(8) * A) * (x + f(1)) - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
    c2 = ((A + f(2)) * (f(1) - x) - (A + f(3))) * (f(1) - x) * (f(1) - x) + f(1)
    c3 = f(1) - c0 - c1 - c2
    return np.array([c0, c1, c2, c3], dtype=np.float32)


def upsample_tables(n_src, ratio):
    """Per destination index: 4 clamped source indices and 4 float32 weights."""
    d = np.arange(n_src * ratio)
    fx = ((d + 0.5) * (1.0 / ratio) - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    frac = fx - s.astype(np.float32)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    w = np.stack([cubic_coeffs(t) for t in frac]).astype(np.float32)
    return idx, w


def upsample_cubic(img, ratio):
    """img (h, w, C) float32 -> (h*ratio, w*ratio, C) float32."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w, _ = img.shape
    xi, xw = upsample_tables(w, ratio)
    yi, yw = upsample_tables(h, ratio)
    t = img[:, xi[:, 0]] * xw[None, :, 0, None]
    for k in (1, 2, 3):
        t = t + img[:, xi[:, k]] * xw[None, :, k, None]
    o = t[yi[:, 0]] * yw[:, 0, None, None]
    for k in (1, 2, 3):
        o = o + t[yi[:, k]] * yw[:, k, None, None]
    return o


def resize_tables(n_src, n_dst):
    """OpenCV generic cubic resize along one axis for an explicit destination size: per destination index 4
    clamped source indices and 4 float32 weights (inv_scale = dst/src, scale = 1/inv_scale, resize.cpp)."""
    inv = float(n_dst) / float(n_src)
    scale = 1.0 / inv
    d = np.arange(n_dst)
    fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(fx).astype(np.int64)
    frac = fx - s.astype(np.float32)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    w = np.stack([cubic_coeffs(t) for t in frac]).astype(np.float32)
    return idx, w


def resize_cubic(img, dst_w, dst_h):
    """cv2.resize(img, (dst_w, dst_h), interpolation=cv2.INTER_CUBIC) for float32 (h, w, C) maps (val.py:100,107):
    horizontal pass then vertical pass, left-to-right float32 sums."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w, _ = img.shape
    xi, xw = resize_tables(w, dst_w)
    yi, yw = resize_tables(h, dst_h)
    t = img[:, xi[:, 0]] * xw[None, :, 0, None]
    for k in (1, 2, 3):
        t = t + img[:, xi[:, k]] * xw[None, :, k, None]
    o = t[yi[:, 0]] * yw[:, 0, None, None]
    for k in (1, 2, 3):
        o = o + t[yi[:, k]] * yw[:, k, None, None]
    return o


def resize_cubic_f64_by_ratio(img, ratio):
    """cv2.resize(img, (0,0), fx=ratio, fy=ratio, INTER_CUBIC) for a float64 image (val.py:89: the normalised image is
    float64): destination size round(src*ratio), scale = 1/ratio, float32 coefficients, float64 accumulation."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    h, w, _ = img.shape
    dw, dh = int(round(w * ratio)), int(round(h * ratio))

    def tables(n_src, n_dst):
        scale = 1.0 / ratio
        d = np.arange(n_dst)
        fx = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(fx).astype(np.int64)
        frac = fx - s.astype(np.float32)
        idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, np.stack([cubic_coeffs(t) for t in frac]).astype(np.float64)
    xi, xw = tables(w, dw)
    yi, yw = tables(h, dh)
    t = img[:, xi[:, 0]] * xw[None, :, 0, None]
    for k in (1, 2, 3):
        t = t + img[:, xi[:, k]] * xw[None, :, k, None]
    o = t[yi[:, 0]] * yw[:, 0, None, None]
    for k in (1, 2, 3):
        o = o + t[yi[:, k]] * yw[:, k, None, None]
    return o


def multiscale_accumulate(avg, maps_chw, stride, pad, width, height, n_scales):
    """One scale of val.infer's inner loop for one map set (val.py:96-101): x`stride` cubic up-sampling, crop of the
    padding, cubic resize to (width, height), avg + maps / n_scales."""
    m = upsample_cubic(np.ascontiguousarray(maps_chw, dtype=np.float32).transpose(1, 2, 0), stride)
    m = m[pad[0]:m.shape[0] - pad[2], pad[1]:m.shape[1] - pad[3], :]
    m = resize_cubic(m, width, height)
    return avg + m / n_scales


# ----------------------------------------------------------------------------- extract
def extract_keypoints(heatmap, all_keypoints, total_keypoint_num):
    heatmap[heatmap < 0.1] = 0          # in place, like the reference (keypoints.py:17)
    h, w = heatmap.shape
    p = np.zeros((h + 2, w + 2), heatmap.dtype)
    p[1:-1, 1:-1] = heatmap
    c = p[1:-1, 1:-1]
    peak = (c > p[1:-1, 2:]) & (c > p[1:-1, :-2]) & (c > p[2:, 1:-1]) & (c > p[:-2, 1:-1])
    ys, xs = np.nonzero(peak)
    order = np.lexsort((ys, xs))        # x ascending, ties y ascending (= stable sort by x of row-major scan)
    xs, ys = xs[order], ys[order]
    n = len(xs)
    alive = np.ones(n, bool)
    found = []
    for i in range(n):
        if not alive[i]:
            continue
        d2 = (xs[i + 1:] - xs[i]) ** 2 + (ys[i + 1:] - ys[i]) ** 2
        alive[i + 1:] &= d2 >= 36       # sqrt(d2) < 6  <=>  d2 < 36 for integers
        found.append((xs[i], ys[i], heatmap[ys[i], xs[i]], total_keypoint_num + len(found)))
    all_keypoints.append(found)
    return len(found)


# ----------------------------------------------------------------------------- group
def score_pairs(ka, kb, paf2, demo, height_n, min_paf_score=0.05, point_num=10, state=None):
    """All (i, j) pairs of one limb -> list of [i, j, ratio, score_all] that pass the
    line-integral test, in (i major, j minor) order (keypoints.py:94-139)."""
    if state is None:
        state = {"ratio_bound": False}
    na, nb = len(ka), len(kb)
    a = np.array([[k[0], k[1]] for k in ka], dtype=np.int64)
    b = np.array([[k[0], k[1]] for k in kb], dtype=np.int64)
    d = b[None, :, :] - a[:, None, :]                            # (na, nb, 2) int64
    norm = np.sqrt((d[..., 0] ** 2 + d[..., 1] ** 2).astype(np.float64))
    valid = norm != 0
    with np.errstate(divide="ignore", invalid="ignore"):
        ux = d[..., 0] / norm
        uy = d[..., 1] / norm
    mx = np.round((a[:, None, 0] + b[None, :, 0]) * 0.5).astype(np.int64)
    my = np.round((a[:, None, 1] + b[None, :, 1]) * 0.5).astype(np.int64)
    mid = ux * paf2[my, mx, 0] + uy * paf2[my, mx, 1]
    mid_ok = mid > -100
    step = (1 / (point_num - 1)) * d                             # float64
    k = np.arange(point_num)
    x = step[..., 0, None] * k + a[:, None, 0, None]
    y = step[..., 1, None] * k + a[:, None, 1, None]
    if demo:
        px, py = np.trunc(x).astype(np.int64), np.trunc(y).astype(np.int64)
    else:
        px, py = np.round(x).astype(np.int64), np.round(y).astype(np.int64)
    acc = np.zeros((na, nb), np.float64)
    cnt = np.zeros((na, nb), np.int64)
    for t in range(point_num):
        sc = ux * paf2[py[..., t], px[..., t], 0] + uy * paf2[py[..., t], px[..., t], 1]
        ok = sc > min_paf_score
        acc = np.where(ok, acc + sc, acc)
        cnt += ok
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(cnt > 0, acc / np.maximum(cnt, 1), 0.0)
        ratio = ratio + np.minimum(height_n / norm - 1, 0)
    out = []
    for i in range(na):
        for j in range(nb):
            if not valid[i, j]:
                continue
            if mid_ok[i, j]:
                state["ratio_bound"] = True
            elif not state["ratio_bound"]:
                # the reference reads `ratio` before any assignment here (keypoints.py:116,137)
                raise UnboundLocalError("local variable 'ratio' referenced before assignment")
            if mid_ok[i, j] and ratio[i, j] > 0 and cnt[i, j] / point_num > 0.8:
                out.append([i, j, ratio[i, j], ratio[i, j] + ka[i][2] + kb[j][2]])
    return out


def _fresh(size):
    return np.ones(size) * -1


def group_keypoints(all_keypoints_by_type, pafs, pose_entry_size=20, min_paf_score=0.05, demo=False):
    entries = []
    all_keypoints = np.array([kp for kps in all_keypoints_by_type for kp in kps])
    height_n = pafs.shape[0] // 2
    state = {"ratio_bound": False}
    for part, ((ta, tb), chans) in enumerate(zip(KPT_IDS, PAF_IDS)):
        ka, kb = all_keypoints_by_type[ta], all_keypoints_by_type[tb]
        na, nb = len(ka), len(kb)
        if na == 0 and nb == 0:
            continue
        if na == 0 or nb == 0:           # one-sided: seed single-keypoint poses (keypoints.py:65-92)
            slot, ks = (tb, kb) if na == 0 else (ta, ka)
            for kp in ks:
                if not any(e[slot] == kp[3] for e in entries):
                    e = _fresh(pose_entry_size)
                    e[slot] = kp[3]
                    e[-1] = 1
                    e[-2] = kp[2]
                    entries.append(e)
            continue

        cand = score_pairs(ka, kb, pafs[:, :, chans], demo, height_n, min_paf_score, 10, state)
        cand.sort(key=lambda c: -c[2])   # stable, descending ratio (keypoints.py:141)
        used_a, used_b = np.zeros(na, bool), np.zeros(nb, bool)
        conns = []
        for i, j, ratio, _ in cand:
            if len(conns) == min(na, nb):
                break
            if not used_a[i] and not used_b[j]:
                conns.append([ka[i][3], kb[j][3], ratio])
                used_a[i] = used_b[j] = True
        if not conns:
            continue

        if part == 0:
            entries = []
            for ia, ib, ratio in conns:
                e = _fresh(pose_entry_size)
                e[ta], e[tb], e[-1] = ia, ib, 2
                e[-2] = np.sum(all_keypoints[[ia, ib], 2]) + ratio
                entries.append(e)
        elif part in (17, 18):
            for ia, ib, _ in conns:
                for e in entries:
                    if e[ta] == ia and e[tb] == -1:
                        e[tb] = ib
                    elif e[tb] == ib and e[ta] == -1:
                        e[ta] = ia
        else:
            for ia, ib, ratio in conns:
                hit = False
                for e in entries:
                    if e[ta] == ia:
                        e[tb] = ib
                        hit = True
                        e[-1] += 1
                        e[-2] += all_keypoints[ib, 2] + ratio
                if not hit:
                    e = _fresh(pose_entry_size)
                    e[ta], e[tb], e[-1] = ia, ib, 2
                    e[-2] = np.sum(all_keypoints[[ia, ib], 2]) + ratio
                    entries.append(e)

    kept = [e for e in entries if not (e[-1] < 3 or e[-2] / e[-1] < 0.2)]
    return np.asarray(kept), all_keypoints
