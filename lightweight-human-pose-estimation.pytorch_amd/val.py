"""Host pre-processing helpers with the reference's names and semantics (reference: val.py:30-49).

OpenCV is not a dependency: ``cv2.copyMakeBorder(BORDER_CONSTANT)`` is a constant fill + copy.  The uint8
``cv2.resize(INTER_CUBIC)`` of demo.py:59 runs on the GPU (``Engine.preprocess_u8``, OpenCV's fixed-point
algorithm: 11-bit coefficients, A = -0.75) — unpinned against cv2 itself, which is absent from the build container.
"""
import math

import numpy as np


def normalize(img, img_mean, img_scale):
    # float32 image minus an integer tuple promotes to float64 under NumPy 2, exactly like the reference
    return (np.array(img, dtype=np.float32) - img_mean) * img_scale


def pad_width(img, stride, pad_value, min_dims):
    """Centre-pad ``img`` to (min_dims rounded up to ``stride``); returns (padded, [top, left, bottom, right]).
    Mutates ``min_dims`` like the reference (val.py:39-41)."""
    h, w, _ = img.shape
    h = min(min_dims[0], h)
    min_dims[0] = math.ceil(min_dims[0] / float(stride)) * stride
    min_dims[1] = math.ceil(max(min_dims[1], w) / float(stride)) * stride
    top = int(math.floor((min_dims[0] - h) / 2.0))
    left = int(math.floor((min_dims[1] - w) / 2.0))
    pad = [top, left, int(min_dims[0] - h - top), int(min_dims[1] - w - left)]
    out = np.empty((img.shape[0] + pad[0] + pad[2], w + pad[1] + pad[3], img.shape[2]), dtype=img.dtype)
    out[...] = np.asarray(pad_value, dtype=img.dtype)[:img.shape[2]]
    out[pad[0]:pad[0] + img.shape[0], pad[1]:pad[1] + w] = img
    return out, pad


def _cubic_coeffs_f32(x):
    f = np.float32
    x = f(x); A = f(-0.75)
    c0 = ((A * (x + f(1)) - f(5) * A) * (x + f(1)) + f(8) * A) * (x + f(1)) - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
    c2 = ((A + f(2)) * (f(1) - x) - (A + f(3))) * (f(1) - x) * (f(1) - x) + f(1)
    return np.array([c0, c1, c2, f(1) - c0 - c1 - c2], dtype=np.float32)


def resize_cubic_float(img, ratio):
    """cv2.resize(img, (0,0), fx=ratio, fy=ratio, interpolation=cv2.INTER_CUBIC) for the float64 normalised image
    (val.py:89): destination size round(src*ratio), scale 1/ratio, float32 coefficients, float64 sums."""
    img = np.ascontiguousarray(img, dtype=np.float64)
    h, w = img.shape[:2]
    dw, dh = int(round(w * ratio)), int(round(h * ratio))

    def tables(n_src, n_dst):
        d = np.arange(n_dst)
        f = ((d + 0.5) * (1.0 / ratio) - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        frac = f - s.astype(np.float32)
        idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
        return idx, np.stack([_cubic_coeffs_f32(t) for t in frac]).astype(np.float64)
    xi, xw = tables(w, dw)
    yi, yw = tables(h, dh)
    t = sum(img[:, xi[:, k]] * xw[None, :, k, None] for k in range(4))
    return sum(t[yi[:, k]] * yw[:, k, None, None] for k in range(4))


def infer(net, img, scales, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1/256):
    """Drop-in for the reference's multi-scale ``val.infer`` (val.py:81-110): returns (avg_heatmaps HxWx19,
    avg_pafs HxWx38) float32 at the original image size.  The image-side resize/pad stay on the host like the
    reference; the network and the per-scale up-sample / crop / resize / average run on the GPU."""
    import torch
    normed_img = normalize(img, img_mean, img_scale)
    height, width, _ = normed_img.shape
    scales_ratios = [scale * base_height / float(height) for scale in scales]
    eng = net.engine
    dev = torch.device("cuda", eng.device_id)
    avg_heatmaps = torch.zeros((height, width, eng.NH), dtype=torch.float32, device=dev)
    avg_pafs = torch.zeros((height, width, eng.NP), dtype=torch.float32, device=dev)
    for ratio in scales_ratios:
        scaled_img = resize_cubic_float(normed_img, ratio)
        min_dims = [base_height, max(scaled_img.shape[1], base_height)]
        padded_img, pad = pad_width(scaled_img, stride, pad_value, min_dims)
        x = torch.from_numpy(np.ascontiguousarray(padded_img.transpose(2, 0, 1)[None], dtype=np.float32)).to(dev)
        stages_output = net(x)
        eng.multiscale_accumulate(avg_heatmaps, stages_output[-2], stride, pad, len(scales_ratios))
        eng.multiscale_accumulate(avg_pafs, stages_output[-1], stride, pad, len(scales_ratios))
    return avg_heatmaps.cpu().numpy(), avg_pafs.cpu().numpy()


def scaled_inputs(imgs, scales, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1/256):
    """Host side of val.py:84-93 for a batch of same-sized frames: per scale the float32 (N,3,H',W') network input
    and its pad [top, left, bottom, right]."""
    height = imgs[0].shape[0]
    out = []
    for ratio in [scale * base_height / float(height) for scale in scales]:
        xs, pad = [], None
        for img in imgs:
            scaled_img = resize_cubic_float(normalize(img, img_mean, img_scale), ratio)
            min_dims = [base_height, max(scaled_img.shape[1], base_height)]
            padded_img, pad = pad_width(scaled_img, stride, pad_value, min_dims)
            xs.append(padded_img.transpose(2, 0, 1))
        out.append((np.ascontiguousarray(np.stack(xs), dtype=np.float32), pad))
    return out


def accumulate_scales(net, inputs, height, width, stride):
    """Device side of val.py:94-108 for a batch: inputs = [(x (N,3,H',W') cuda tensor, pad), ...] -> averaged maps
    (N,height,width,19) and (N,height,width,38) float32 cuda tensors."""
    import torch
    eng = net.engine
    dev = torch.device("cuda", eng.device_id)
    N = int(inputs[0][0].shape[0])
    avg_heatmaps = torch.empty((N, height, width, eng.NH), dtype=torch.float32, device=dev)   # zero-initialised by the first scale
    avg_pafs = torch.empty((N, height, width, eng.NP), dtype=torch.float32, device=dev)
    for k, (x, pad) in enumerate(inputs):
        stages_output = net(x)
        eng.multiscale_accumulate(avg_heatmaps, stages_output[-2], stride, pad, len(inputs), init=(k == 0))
        eng.multiscale_accumulate(avg_pafs, stages_output[-1], stride, pad, len(inputs), init=(k == 0))
    return avg_heatmaps, avg_pafs


def infer_batch(net, imgs, scales, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1/256):
    """``infer`` (val.py:81-110) for N same-sized frames at once; returns cuda tensors (N,H,W,19), (N,H,W,38)."""
    import torch
    dev = torch.device("cuda", net.engine.device_id)
    inputs = [(torch.from_numpy(x).to(dev), pad) for x, pad in scaled_inputs(imgs, scales, base_height, stride, pad_value, img_mean, img_scale)]
    return accumulate_scales(net, inputs, imgs[0].shape[0], imgs[0].shape[1], stride)


def poses_batch(net, avg_heatmaps, avg_pafs):
    """val.py:129-134 for a batch on the device: extract_keypoints over the 18 key-point maps + group_keypoints
    (demo=False rounding) -> per frame (pose_entries (P,20), all_keypoints (K,4), type_counts)."""
    return net.engine.poses_from_maps(avg_heatmaps, avg_pafs, 1, demo=False, layout="NHWC")


# ---------------------------------------------------------------------------------------------- COCO results
# slot of each of the 18 network key-points in COCO's 17-key-point order (the neck, index 1, has none)
_COCO_SLOT = (0, None, 6, 8, 10, 5, 7, 9, 12, 14, 16, 11, 13, 15, 2, 1, 4, 3)


def convert_to_coco_format(pose_entries, all_keypoints):
    """Same contract as the reference's val.convert_to_coco_format (val.py:52-78): per pose a flat list of 17 x
    (x + 0.5, y + 0.5, visibility) in COCO order (missing key-points stay 0, 0, 0) and the pose score
    ``entry[18] * max(0, entry[19] - 1)`` (the neck does not count)."""
    coco_keypoints, scores = [], []
    for entry in pose_entries:
        if len(entry) == 0:
            continue
        flat = [0] * (17 * 3)
        for part, slot in enumerate(_COCO_SLOT):
            if slot is None:
                continue
            kpt_id = entry[part]
            if kpt_id != -1:
                cx, cy = all_keypoints[int(kpt_id), 0:2]
                flat[slot * 3:slot * 3 + 3] = [cx + 0.5, cy + 0.5, 1]
        coco_keypoints.append(flat)
        scores.append(entry[-2] * max(0, entry[-1] - 1))
    return coco_keypoints, scores


def coco_detections(net, samples, multiscale=False, base_height=368, stride=8):
    """The detection loop of val.evaluate (val.py:113-147) without the dataset / pycocotools parts: ``samples`` yields
    dicts with 'file_name' ('<image id>.jpg') and 'img' (uint8 HxWx3); returns the list of COCO result dicts."""
    scales = [0.5, 1.0, 1.5, 2.0] if multiscale else [1]
    results = []
    for sample in samples:
        file_name, img = sample['file_name'], sample['img']
        avg_heatmaps, avg_pafs = infer_batch(net, [img], scales, base_height, stride)
        pose_entries, all_keypoints, _ = poses_batch(net, avg_heatmaps, avg_pafs)[0]
        coco_keypoints, scores = convert_to_coco_format(pose_entries, all_keypoints)
        image_id = int(file_name[0:file_name.rfind('.')])
        for keypoints, score in zip(coco_keypoints, scores):
            results.append({'image_id': image_id, 'category_id': 1, 'keypoints': [float(v) for v in keypoints], 'score': float(score)})
    return results


def write_detections(output_name, coco_result):
    import json
    with open(output_name, 'w') as f:
        json.dump(coco_result, f, indent=4)
