"""ORACLE — test infrastructure, NOT product code.

NumPy restatement of the reference's host pre-processing and of the multi-scale driver:
  val.py:30-33   normalize   ((img - mean) * scale; float64 under NumPy 2)
  val.py:36-49   pad_width   (centre pad to a multiple of the stride; pad = [top, left, bottom, right])
  demo.py:55-64  prepare_frame (uint8 cubic resize by height, normalize, pad, HWC -> 1x3xHxW float32)
  val.py:81-110  infer       (per scale: cubic resize of the normalised image, pad, net, x8 up-sample, crop,
                              resize to the original size, running average)
cv2 is absent from the build container: the resizes follow OpenCV's published cubic algorithm
(oracle/post_ref.py) — "parity unpinned vs cv2".
"""
import math

import numpy as np
import torch

from . import net_ref, post_ref


def normalize(img, img_mean, img_scale):
    return (np.array(img, dtype=np.float32) - img_mean) * img_scale


def pad_width(img, stride, pad_value, min_dims):
    h, w, _ = img.shape
    h = min(min_dims[0], h)
    H = math.ceil(min_dims[0] / float(stride)) * stride
    W = math.ceil(max(min_dims[1], w) / float(stride)) * stride
    top, left = int(math.floor((H - h) / 2.0)), int(math.floor((W - w) / 2.0))
    pad = [top, left, int(H - h - top), int(W - w - left)]
    out = np.zeros((img.shape[0] + pad[0] + pad[2], w + pad[1] + pad[3], img.shape[2]), img.dtype)
    out[...] = np.asarray(pad_value, dtype=img.dtype)
    out[pad[0]:pad[0] + img.shape[0], pad[1]:pad[1] + w] = img
    return out, pad


def _cubic_coeffs_f32(x):
    f = np.float32
    x = f(x); A = f(-0.75)
    c0 = ((A * (x + f(1)) - f(5) * A) * (x + f(1)) + f(8) * A) * (x + f(1)) - f(4) * A
    c1 = ((A + f(2)) * x - (A + f(3))) * x * x + f(1)
    c2 = ((A + f(2)) * (f(1) - x) - (A + f(3))) * (f(1) - x) * (f(1) - x) + f(1)
    return np.array([c0, c1, c2, f(1) - c0 - c1 - c2], dtype=np.float32)


def _axis_tables_u8(n_src, n_dst, inv_scale):
    d = np.arange(n_dst)
    f = ((d + 0.5) * (1.0 / inv_scale) - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    frac = f - s.astype(np.float32)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    w = np.stack([_cubic_coeffs_f32(t) for t in frac]) * np.float32(2048)
    return idx, np.clip(np.rint(w), -32768, 32767).astype(np.int64)


def resize_cubic_u8(img, fx, fy):
    """cv2.resize(img, (0,0), fx=fx, fy=fy, interpolation=cv2.INTER_CUBIC) for uint8 HxWxC (demo.py:59), restated from
    OpenCV's fixed-point path: float32 coefficients (A = -0.75) * 2048 rounded to short, integer horizontal and
    vertical sums, (v + 2^21) >> 22, saturate.  Unpinned vs cv2 (absent here; its SIMD vertical pass may differ by 1)."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    dw, dh = int(round(w * fx)), int(round(h * fy))
    xi, xw = _axis_tables_u8(w, dw, fx)
    yi, yw = _axis_tables_u8(h, dh, fy)
    src = img.astype(np.int64)
    t = sum(src[:, xi[:, k]] * xw[None, :, k, None] for k in range(4))
    o = sum(t[yi[:, k]] * yw[:, k, None, None] for k in range(4))
    return np.clip((o + (1 << 21)) >> 22, 0, 255).astype(np.uint8)


def prepare_frame(img, net_input_height_size, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1 / 256):
    """demo.py:55-64: returns (x 1x3xH'xW' float32, scale, pad)."""
    height = img.shape[0]
    scale = net_input_height_size / height
    scaled = resize_cubic_u8(img, scale, scale)
    scaled = normalize(scaled, img_mean, img_scale)
    min_dims = [net_input_height_size, max(scaled.shape[1], net_input_height_size)]
    padded, pad = pad_width(scaled, stride, pad_value, min_dims)
    x = np.ascontiguousarray(padded.transpose(2, 0, 1)[None]).astype(np.float32)
    return x, scale, pad


def infer(sd, nref, img, scales, base_height, stride, pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1 / 256):
    normed = normalize(img, img_mean, img_scale)
    height, width, _ = normed.shape
    ratios = [s * base_height / float(height) for s in scales]
    avg_h = np.zeros((height, width, 19), np.float32)
    avg_p = np.zeros((height, width, 38), np.float32)
    for r in ratios:
        scaled = post_ref.resize_cubic_f64_by_ratio(normed, r)
        padded, pad = pad_width(scaled, stride, pad_value, [base_height, max(scaled.shape[1], base_height)])
        x = torch.from_numpy(np.ascontiguousarray(padded.transpose(2, 0, 1)[None], dtype=np.float32))
        outs = net_ref.forward(sd, x, nref)
        avg_h = post_ref.multiscale_accumulate(avg_h, outs[-2][0].numpy(), stride, pad, width, height, len(ratios))
        avg_p = post_ref.multiscale_accumulate(avg_p, outs[-1][0].numpy(), stride, pad, width, height, len(ratios))
    return avg_h, avg_p
