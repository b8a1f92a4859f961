"""Pose container, similarity and id tracking (reference: modules/pose.py:8-118), without OpenCV:
``cv2.boundingRect`` of integer points is (min_x, min_y, max_x-min_x+1, max_y-min_y+1)."""
import numpy as np

from .keypoints import BODY_PARTS_KPT_IDS, BODY_PARTS_PAF_IDS
from .one_euro_filter import OneEuroFilter


class Pose:
    num_kpts = 18
    kpt_names = ['nose', 'neck', 'r_sho', 'r_elb', 'r_wri', 'l_sho', 'l_elb', 'l_wri', 'r_hip', 'r_knee', 'r_ank',
                 'l_hip', 'l_knee', 'l_ank', 'r_eye', 'l_eye', 'r_ear', 'l_ear']
    sigmas = np.array([.26, .79, .79, .72, .62, .79, .72, .62, 1.07, .87, .89, 1.07, .87, .89, .25, .25, .35, .35],
                      dtype=np.float32) / 10.0
    vars = (sigmas * 2) ** 2
    last_id = -1
    color = [0, 224, 255]

    def __init__(self, keypoints, confidence):
        self.keypoints = keypoints
        self.confidence = confidence
        self.bbox = Pose.get_bbox(self.keypoints)
        self.id = None
        self.filters = [[OneEuroFilter(), OneEuroFilter()] for _ in range(Pose.num_kpts)]

    @staticmethod
    def get_bbox(keypoints):
        found = keypoints[keypoints[:, 0] != -1]
        if len(found) == 0:
            return (0, 0, 0, 0)
        x0, y0 = int(found[:, 0].min()), int(found[:, 1].min())
        return (x0, y0, int(found[:, 0].max()) - x0 + 1, int(found[:, 1].max()) - y0 + 1)

    def update_id(self, id=None):
        self.id = id
        if self.id is None:
            self.id = Pose.last_id + 1
            Pose.last_id += 1

    def draw(self, img):
        """Minimal rasteriser (filled 3-px discs and 2-px segments) — OpenCV is not a dependency here."""
        assert self.keypoints.shape == (Pose.num_kpts, 2)
        h, w = img.shape[:2]

        def disc(cx, cy, r=3):
            y0, y1, x0, x1 = max(cy - r, 0), min(cy + r + 1, h), max(cx - r, 0), min(cx + r + 1, w)
            if y0 >= y1 or x0 >= x1:
                return
            yy, xx = np.mgrid[y0:y1, x0:x1]
            img[y0:y1, x0:x1][(yy - cy) ** 2 + (xx - cx) ** 2 <= r * r] = Pose.color

        for part_id in range(len(BODY_PARTS_PAF_IDS) - 2):
            a, b = BODY_PARTS_KPT_IDS[part_id]
            has_a, has_b = self.keypoints[a, 0] != -1, self.keypoints[b, 0] != -1
            if has_a:
                disc(int(self.keypoints[a, 0]), int(self.keypoints[a, 1]))
            if has_b:
                disc(int(self.keypoints[b, 0]), int(self.keypoints[b, 1]))
            if has_a and has_b:
                n = int(max(abs(self.keypoints[a] - self.keypoints[b]).max(), 1))
                for t in np.linspace(0.0, 1.0, n + 1):
                    p = self.keypoints[a] + t * (self.keypoints[b] - self.keypoints[a])
                    disc(int(p[0]), int(p[1]), 1)


def get_similarity(a, b, threshold=0.5):
    num_similar_kpt = 0
    for kpt_id in range(Pose.num_kpts):
        if a.keypoints[kpt_id, 0] != -1 and b.keypoints[kpt_id, 0] != -1:
            distance = np.sum((a.keypoints[kpt_id] - b.keypoints[kpt_id]) ** 2)
            area = max(a.bbox[2] * a.bbox[3], b.bbox[2] * b.bbox[3])
            similarity = np.exp(-distance / (2 * (area + np.spacing(1)) * Pose.vars[kpt_id]))
            if similarity > threshold:
                num_similar_kpt += 1
    return num_similar_kpt


def track_poses(previous_poses, current_poses, threshold=3, smooth=False):
    """Propagate ids from the previous frame (>= ``threshold`` similar key-points), optionally smoothing."""
    current_poses = sorted(current_poses, key=lambda pose: pose.confidence, reverse=True)
    free = np.ones(len(previous_poses), dtype=np.int32)
    for cur in current_poses:
        best_idx, best_pose_id, best_sim = None, None, 0
        for idx, prev in enumerate(previous_poses):
            if not free[idx]:
                continue
            sim = get_similarity(cur, prev)
            if sim > best_sim:
                best_sim, best_pose_id, best_idx = sim, prev.id, idx
        if best_sim >= threshold:
            free[best_idx] = 0
        else:
            best_pose_id = None
        cur.update_id(best_pose_id)
        if smooth:
            for k in range(Pose.num_kpts):
                if cur.keypoints[k, 0] == -1:
                    continue
                if best_pose_id is not None and previous_poses[best_idx].keypoints[k, 0] != -1:
                    cur.filters[k] = previous_poses[best_idx].filters[k]
                cur.keypoints[k, 0] = cur.filters[k][0](cur.keypoints[k, 0])
                cur.keypoints[k, 1] = cur.filters[k][1](cur.keypoints[k, 1])
            cur.bbox = Pose.get_bbox(cur.keypoints)
