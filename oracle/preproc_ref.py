"""ORACLE — test infrastructure, NOT product code.

NumPy restatement of the reference's host pre-processing and of the multi-scale driver:
  val.py:30-33   normalize   ((img - mean) * scale; float64 under NumPy 2)
  val.py:36-49   pad_width   (centre pad to a multiple of the stride; pad = [top, left, bottom, right])
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
