"""ORACLE — test infrastructure, NOT product code.

Plain-Python restatement of the tail of the reference's hot path, kept independent of the product package:
  demo.py:101-103            un-map of the key-point coordinates: x = (x * stride / upsample_ratio - pad[1]) / scale,
                             y likewise with pad[0]  (pad = [top, left, bottom, right])
  demo.py:104-114            per pose entry: (18, 2) int32 key-points, -1 = missing, ``int()`` truncation toward zero,
                             confidence = entry[18]
  modules/pose.py:21-39      Pose record; bbox = cv2.boundingRect(found key-points)
                             = (min_x, min_y, max_x - min_x + 1, max_y - min_y + 1)   ["parity unpinned vs cv2": OpenCV is
                             absent from the build container; the formula is OpenCV's documented integer-point rule]
  modules/pose.py:41-45      update_id (class-level running id)
  modules/pose.py:65-75      get_similarity
  modules/pose.py:78-118     track_poses (confidence-descending greedy matching, mask, filter hand-over, bbox refresh)
  modules/one_euro_filter.py:4-43  the smoothing filter used by track_poses(smooth=True)

Pinning: the 1-Euro sequence is checked against tests/golden/one_euro.json and get_similarity / track_poses / update_id
against tests/golden/tracking.json — both captured from the REFERENCE's own function bodies by oracle/make_golden.py
(modules/pose.py imports cv2, so its plain-NumPy definitions are parsed out of the file and executed unchanged; the only
stand-in is cv2.boundingRect).  The un-map / Pose construction (demo.py imports cv2) is pinned by hand-derived cases in
tests/test_host_logic.py; boundingRect itself stays "parity unpinned vs cv2".
"""
import math

import numpy as np

NUM_KPTS = 18
_SIGMAS = np.array([.26, .79, .79, .72, .62, .79, .72, .62, 1.07, .87, .89, 1.07, .87, .89, .25, .25, .35, .35],
                   dtype=np.float32) / 10.0
VARS = (_SIGMAS * 2) ** 2


# ----------------------------------------------------------------------------- one_euro_filter.py:4-43
def alpha_of(rate, cutoff):
    tau = 1 / (2 * math.pi * cutoff)
    te = 1 / rate
    return 1 / (1 + tau / te)


class RefLowPass(object):
    def __init__(self):
        self.prev = None

    def step(self, x, alpha):
        if self.prev is None:
            self.prev = x
            return x
        y = alpha * x + (1 - alpha) * self.prev
        self.prev = y
        return y


class RefOneEuro(object):
    def __init__(self, freq=15, mincutoff=1, beta=0.05, dcutoff=1):
        self.freq, self.mincutoff, self.beta, self.dcutoff = freq, mincutoff, beta, dcutoff
        self.lp_x, self.lp_dx = RefLowPass(), RefLowPass()
        self.raw_prev = None
        self.dx = None

    def __call__(self, x):
        if self.dx is None:
            self.dx = 0
        else:
            self.dx = (x - self.raw_prev) * self.freq
        dx_hat = self.lp_dx.step(self.dx, alpha_of(self.freq, self.dcutoff))
        cutoff = self.mincutoff + self.beta * abs(dx_hat)
        y = self.lp_x.step(x, alpha_of(self.freq, cutoff))
        self.raw_prev = x
        return y


# ----------------------------------------------------------------------------- pose.py:21-45
def bounding_rect(keypoints):
    """cv2.boundingRect of the found (!= -1) integer points; no point found: OpenCV returns the empty rect (0, 0, 0, 0)."""
    xs = [int(keypoints[k, 0]) for k in range(NUM_KPTS) if keypoints[k, 0] != -1]
    ys = [int(keypoints[k, 1]) for k in range(NUM_KPTS) if keypoints[k, 0] != -1]
    if not xs:
        return (0, 0, 0, 0)
    return (min(xs), min(ys), max(xs) - min(xs) + 1, max(ys) - min(ys) + 1)


class RefPose(object):
    last_id = -1

    def __init__(self, keypoints, confidence):
        self.keypoints = keypoints
        self.confidence = confidence
        self.bbox = bounding_rect(keypoints)
        self.id = None
        self.filters = [[RefOneEuro(), RefOneEuro()] for _ in range(NUM_KPTS)]

    def update_id(self, id=None):
        self.id = id
        if self.id is None:
            self.id = RefPose.last_id + 1
            RefPose.last_id += 1


# ----------------------------------------------------------------------------- demo.py:101-114
def poses_from_entries(pose_entries, all_keypoints, scale, pad, stride=8, upsample_ratio=4):
    """Returns the list of RefPose built from group_keypoints' outputs; ``all_keypoints`` is modified in place like the
    reference does (demo.py:102-103)."""
    for kpt_id in range(all_keypoints.shape[0] if all_keypoints.ndim == 2 else 0):
        all_keypoints[kpt_id, 0] = (all_keypoints[kpt_id, 0] * stride / upsample_ratio - pad[1]) / scale
        all_keypoints[kpt_id, 1] = (all_keypoints[kpt_id, 1] * stride / upsample_ratio - pad[0]) / scale
    poses = []
    for n in range(len(pose_entries)):
        if len(pose_entries[n]) == 0:
            continue
        kp = np.ones((NUM_KPTS, 2), dtype=np.int32) * -1
        for kpt_id in range(NUM_KPTS):
            if pose_entries[n][kpt_id] != -1.0:
                kp[kpt_id, 0] = int(all_keypoints[int(pose_entries[n][kpt_id]), 0])
                kp[kpt_id, 1] = int(all_keypoints[int(pose_entries[n][kpt_id]), 1])
        poses.append(RefPose(kp, pose_entries[n][18]))
    return poses


# ----------------------------------------------------------------------------- pose.py:65-118
def get_similarity(a, b, threshold=0.5):
    num = 0
    for kpt_id in range(NUM_KPTS):
        if a.keypoints[kpt_id, 0] != -1 and b.keypoints[kpt_id, 0] != -1:
            distance = np.sum((a.keypoints[kpt_id] - b.keypoints[kpt_id]) ** 2)
            area = max(a.bbox[2] * a.bbox[3], b.bbox[2] * b.bbox[3])
            similarity = np.exp(-distance / (2 * (area + np.spacing(1)) * VARS[kpt_id]))
            if similarity > threshold:
                num += 1
    return num


def track_poses(previous_poses, current_poses, threshold=3, smooth=False):
    current_poses = sorted(current_poses, key=lambda pose: pose.confidence, reverse=True)
    mask = np.ones(len(previous_poses), dtype=np.int32)
    for cur in current_poses:
        best_idx, best_pose_id, best_iou = None, None, 0
        for idx, prev in enumerate(previous_poses):
            if not mask[idx]:
                continue
            iou = get_similarity(cur, prev)
            if iou > best_iou:
                best_iou, best_pose_id, best_idx = iou, prev.id, idx
        if best_iou >= threshold:
            mask[best_idx] = 0          # best_idx None (threshold <= 0, nothing similar): NumPy clears the whole mask
        else:
            best_pose_id = None
        cur.update_id(best_pose_id)
        if smooth:
            for kpt_id in range(NUM_KPTS):
                if cur.keypoints[kpt_id, 0] == -1:
                    continue
                if best_pose_id is not None and previous_poses[best_idx].keypoints[kpt_id, 0] != -1:
                    cur.filters[kpt_id] = previous_poses[best_idx].filters[kpt_id]
                cur.keypoints[kpt_id, 0] = cur.filters[kpt_id][0](cur.keypoints[kpt_id, 0])
                cur.keypoints[kpt_id, 1] = cur.filters[kpt_id][1](cur.keypoints[kpt_id, 1])
            cur.bbox = bounding_rect(cur.keypoints)
