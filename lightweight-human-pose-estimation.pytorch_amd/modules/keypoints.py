"""Drop-ins for the reference's ``modules.keypoints`` (modules/keypoints.py:5-201), running on the GPU.

``extract_keypoints`` and ``group_keypoints`` keep the reference's signatures and conventions:
in-place thresholding of the heat-map, append-to-caller's-list out-parameter, tuples of
(np.int64 x, np.int64 y, np.float32 score, int id), float64 (P,20)/(K,4) results, ``(0,)``-shaped
empties.  The computation is HIP kernels behind the C ABI (lwp_extract_keypoints / lwp_group_keypoints).
"""
import numpy as np

from ..runtime import default_engine

BODY_PARTS_KPT_IDS = [[1, 2], [1, 5], [2, 3], [3, 4], [5, 6], [6, 7], [1, 8], [8, 9], [9, 10], [1, 11],
                      [11, 12], [12, 13], [1, 0], [0, 14], [14, 16], [0, 15], [15, 17], [2, 16], [5, 17]]
BODY_PARTS_PAF_IDS = ([12, 13], [20, 21], [14, 15], [16, 17], [22, 23], [24, 25], [0, 1], [2, 3], [4, 5],
                      [6, 7], [8, 9], [10, 11], [28, 29], [30, 31], [34, 35], [32, 33], [36, 37], [18, 19], [26, 27])


def extract_keypoints(heatmap, all_keypoints, total_keypoint_num, engine=None):
    eng = engine or default_engine()
    xs, ys, sc = eng.extract_keypoints(heatmap)
    found = [(xs[i], ys[i], sc[i], total_keypoint_num + i) for i in range(len(xs))]
    all_keypoints.append(found)
    return len(found)


def group_keypoints(all_keypoints_by_type, pafs, pose_entry_size=20, min_paf_score=0.05, demo=False, engine=None):
    if pose_entry_size != 20 or min_paf_score != 0.05:
        raise ValueError("the HIP path implements the reference defaults pose_entry_size=20, min_paf_score=0.05")
    if len(all_keypoints_by_type) != 18:
        raise ValueError("expected 18 key-point types")
    eng = engine or default_engine()
    all_keypoints = np.array([item for sublist in all_keypoints_by_type for item in sublist])
    counts = np.array([len(s) for s in all_keypoints_by_type], dtype=np.int32)
    kp = all_keypoints.reshape(-1, 4) if all_keypoints.size else np.zeros((0, 4))
    entries = eng.group_keypoints(kp, counts, pafs, demo)
    pose_entries = entries if len(entries) else np.asarray([])
    return pose_entries, all_keypoints
