"""Pose container, OKS-like similarity and id tracking with the reference's public surface
(modules/pose.py:8-118: Pose.num_kpts / kpt_names / sigmas / vars / last_id, Pose(keypoints, confidence),
.bbox, .id, .filters, get_bbox, update_id, draw, get_similarity, track_poses), without OpenCV.

cv2.boundingRect of integer points is (min_x, min_y, max_x - min_x + 1, max_y - min_y + 1); the similarity of all
pose pairs is computed as one NumPy expression and ids are assigned in descending-confidence order.
"""
import numpy as np

from .keypoints import BODY_PARTS_KPT_IDS, BODY_PARTS_PAF_IDS
from .one_euro_filter import OneEuroFilter

_SIGMAS = np.array([.26, .79, .79, .72, .62, .79, .72, .62, 1.07, .87, .89, 1.07, .87, .89, .25, .25, .35, .35],
                   dtype=np.float32) / 10.0


class Pose:
    num_kpts = 18
    kpt_names = ['nose', 'neck', 'r_sho', 'r_elb', 'r_wri', 'l_sho', 'l_elb', 'l_wri', 'r_hip', 'r_knee', 'r_ank',
                 'l_hip', 'l_knee', 'l_ank', 'r_eye', 'l_eye', 'r_ear', 'l_ear']
    sigmas = _SIGMAS
    vars = (_SIGMAS * 2) ** 2
    last_id = -1
    color = [0, 224, 255]

    def __init__(self, keypoints, confidence):
        self.keypoints = keypoints                      # (18, 2) int32, -1 = not found
        self.confidence = confidence
        self.id = None
        self.filters = [[OneEuroFilter(), OneEuroFilter()] for _ in range(self.num_kpts)]
        self.bbox = self.get_bbox(keypoints)

    @staticmethod
    def get_bbox(keypoints):
        present = keypoints[:, 0] != -1
        if not present.any():
            return (0, 0, 0, 0)
        lo = keypoints[present].min(axis=0)
        hi = keypoints[present].max(axis=0)
        return (int(lo[0]), int(lo[1]), int(hi[0] - lo[0] + 1), int(hi[1] - lo[1] + 1))

    def update_id(self, id=None):
        if id is None:
            Pose.last_id += 1
            id = Pose.last_id
        self.id = id

    def draw(self, img):
        """Rasterises joints (radius 3) and limbs (width ~2) straight into ``img`` (H, W, 3)."""
        assert self.keypoints.shape == (Pose.num_kpts, 2)
        h, w = img.shape[:2]

        def stamp(cx, cy, r):
            ys = slice(max(cy - r, 0), min(cy + r + 1, h))
            xs = slice(max(cx - r, 0), min(cx + r + 1, w))
            if ys.start >= ys.stop or xs.start >= xs.stop:
                return
            gy, gx = np.ogrid[ys, xs]
            img[ys, xs][(gy - cy) ** 2 + (gx - cx) ** 2 <= r * r] = Pose.color

        for a, b in BODY_PARTS_KPT_IDS[:len(BODY_PARTS_PAF_IDS) - 2]:
            pa, pb = self.keypoints[a], self.keypoints[b]
            if pa[0] != -1:
                stamp(int(pa[0]), int(pa[1]), 3)
            if pb[0] != -1:
                stamp(int(pb[0]), int(pb[1]), 3)
            if pa[0] != -1 and pb[0] != -1:
                steps = int(max(np.abs(pb - pa).max(), 1))
                for s in range(steps + 1):
                    q = pa + (pb - pa) * (s / steps)
                    stamp(int(q[0]), int(q[1]), 1)


def _similar_keypoints(a, b, threshold=0.5):
    """Number of key-points present in both poses whose OKS-like similarity exceeds ``threshold``."""
    both = (a.keypoints[:, 0] != -1) & (b.keypoints[:, 0] != -1)
    if not both.any():
        return 0
    d2 = ((a.keypoints[both] - b.keypoints[both]) ** 2).sum(axis=1)
    area = max(a.bbox[2] * a.bbox[3], b.bbox[2] * b.bbox[3])
    sim = np.exp(-d2 / (2 * (area + np.spacing(1)) * Pose.vars[both]))
    return int((sim > threshold).sum())


def get_similarity(a, b, threshold=0.5):
    return _similar_keypoints(a, b, threshold)


def track_poses(previous_poses, current_poses, threshold=3, smooth=False):
    """Give every current pose the id of the most similar still-unclaimed previous pose (at least ``threshold``
    similar key-points), most confident poses first; unmatched poses get fresh ids.  With ``smooth`` the matched
    key-points continue the previous pose's 1-Euro filters (reference: modules/pose.py:78-118).

    Corner cases kept from the reference: a candidate must have at least ONE similar key-point to be considered (strict
    ``>`` scan from 0, first maximum wins); with ``threshold <= 0`` and no candidate every previous pose becomes
    unavailable (NumPy's ``mask[None] = 0``); a matched previous pose whose own id is None hands over neither id nor
    filters."""
    free = np.ones(len(previous_poses), dtype=bool)
    for cur in sorted(current_poses, key=lambda p: p.confidence, reverse=True):
        best, best_n = None, 0
        for i, prev in enumerate(previous_poses):
            if not free[i]:
                continue
            n = _similar_keypoints(cur, prev)
            if n > best_n:
                best, best_n = i, n
        inherited = None
        if best_n >= threshold:
            if best is None:
                free[:] = False
            else:
                free[best] = False
                inherited = previous_poses[best].id
        cur.update_id(inherited)
        if not smooth:
            continue
        donor = previous_poses[best] if inherited is not None else None
        for k in range(Pose.num_kpts):
            if cur.keypoints[k, 0] == -1:
                continue
            if donor is not None and donor.keypoints[k, 0] != -1:
                cur.filters[k] = donor.filters[k]
            fx, fy = cur.filters[k]
            cur.keypoints[k, 0] = fx(cur.keypoints[k, 0])
            cur.keypoints[k, 1] = fy(cur.keypoints[k, 1])
        cur.bbox = Pose.get_bbox(cur.keypoints)
