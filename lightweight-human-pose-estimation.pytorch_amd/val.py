"""Drop-ins for the reference's ``val`` helpers (reference: val.py:30-147): ``normalize`` / ``pad_width`` (host, for callers
that use them directly), the multi-scale driver ``infer`` (all of it on the GPU), ``convert_to_coco_format`` and the
detection loop.

OpenCV is not a dependency: ``cv2.copyMakeBorder(BORDER_CONSTANT)`` is a constant fill + copy.  The cubic resizes
(uint8 fixed point of demo.py:59, float64 of val.py:89, float32 of val.py:98-107) run on the GPU with OpenCV's published
algorithms — unpinned against cv2 itself, which is absent from the build container.
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


def _as_frames(imgs):
    """(N, H, W, 3) frames for the device kernels: uint8 stays uint8; any other dtype is cast to float32, which is what the
    reference's normalize does first (np.array(img, dtype=np.float32), val.py:31)."""
    if getattr(imgs, "is_cuda", False):
        import torch
        a = imgs if imgs.dtype in (torch.uint8, torch.float32) else imgs.to(torch.float32)
    else:
        a = np.ascontiguousarray(imgs)
        if a.dtype != np.uint8:
            a = np.ascontiguousarray(a, dtype=np.float32)
    if len(a.shape) != 4 or a.shape[-1] != 3:
        raise TypeError("frames must be (N, H, W, 3)")
    return a


def infer(net, img, scales, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1/256):
    """Drop-in for the reference's multi-scale ``val.infer`` (val.py:81-110): returns (avg_heatmaps HxWx19,
    avg_pafs HxWx38) float32 at the original image size.  Everything between the uint8 frame and the averaged maps runs
    on the GPU: normalize + per-scale cubic resize + pad (lwp_preprocess_scaled_u8), the network, the per-scale
    up-sample / crop / resize / average (lwp_multiscale_accumulate)."""
    avg_heatmaps, avg_pafs = infer_batch(net, np.asarray(img)[None], scales, base_height, stride, pad_value, img_mean, img_scale)
    return avg_heatmaps[0].cpu().numpy(), avg_pafs[0].cpu().numpy()


def scaled_inputs(net, imgs, scales, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1/256):
    """Image side of val.py:84-93 for a batch of same-sized frames (numpy or cuda tensor, N x H x W x 3; uint8, or any other
    dtype through float32 like the reference's normalize), on the device:
    per scale the float32 (N,3,H',W') network input (cuda tensor) and its pad [top, left, bottom, right]."""
    imgs = _as_frames(imgs)
    height = int(imgs.shape[1])
    return [net.engine.preprocess_scaled_u8(imgs, scale * base_height / float(height), base_height, stride, pad_value, img_mean, img_scale)
            for scale in scales]


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
    """``infer`` (val.py:81-110) for N same-sized uint8 frames at once; returns cuda tensors (N,H,W,19), (N,H,W,38)."""
    imgs = _as_frames(np.stack(list(imgs)) if isinstance(imgs, (list, tuple)) else imgs)
    inputs = scaled_inputs(net, imgs, scales, base_height, stride, pad_value, img_mean, img_scale)
    return accumulate_scales(net, inputs, int(imgs.shape[1]), int(imgs.shape[2]), stride)


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
