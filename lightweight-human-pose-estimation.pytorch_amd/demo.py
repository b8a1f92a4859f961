"""Drop-ins for the reference's ``demo.infer_fast`` / ``demo.run_demo`` (reference: demo.py:54-136).

``infer_fast`` keeps the reference signature and return convention (heatmaps HxWx19 f32, pafs HxWx38 f32,
scale, pad); the network and the x``upsample_ratio`` bicubic up-sampling run on the GPU.
``run_demo`` runs the same per-frame pipeline without a GUI and yields the poses; with ``fused=True`` it
uses the single fused C-ABI call (no up-sampled maps are materialised).
"""
import numpy as np

from .modules.keypoints import extract_keypoints, group_keypoints
from .modules.pose import Pose, track_poses


def _prepare(net, img, net_input_height_size, stride, pad_value, img_mean, img_scale):
    """demo.py:55-64 (resize by height, normalize, pad, to tensor) as one GPU kernel: uint8 frame -> cuda tensor.  The tensor never
    leaves infer_fast / run_demo, so it stays on the engine's stream (no hand-over to torch's current stream)."""
    return net.engine.preprocess_u8(img, net_input_height_size, stride, pad_value, img_mean, img_scale, hand_over=False)


def infer_fast(net, img, net_input_height_size, stride, upsample_ratio, cpu,
               pad_value=(0, 0, 0), img_mean=(128, 128, 128), img_scale=1/256):
    x, scale, pad = _prepare(net, img, net_input_height_size, stride, pad_value, img_mean, img_scale)
    stages_output = net(x)                       # numpy in -> numpy out; `cpu` is accepted and ignored
    eng = net.engine
    heatmaps = eng.upsample(stages_output[-2], upsample_ratio)[0]
    pafs = eng.upsample(stages_output[-1], upsample_ratio)[0]
    return heatmaps, pafs, scale, pad


def poses_from_entries(pose_entries, all_keypoints, scale, pad, stride=8, upsample_ratio=4):
    """demo.py:101-114: map key-points back to image coordinates and build Pose objects."""
    num_keypoints = Pose.num_kpts
    all_keypoints = np.array(all_keypoints, dtype=np.float64, copy=True).reshape(-1, 4)
    for kpt_id in range(all_keypoints.shape[0]):
        all_keypoints[kpt_id, 0] = (all_keypoints[kpt_id, 0] * stride / upsample_ratio - pad[1]) / scale
        all_keypoints[kpt_id, 1] = (all_keypoints[kpt_id, 1] * stride / upsample_ratio - pad[0]) / scale
    poses = []
    for entry in pose_entries:
        if len(entry) == 0:
            continue
        kp = np.ones((num_keypoints, 2), dtype=np.int32) * -1
        for kpt_id in range(num_keypoints):
            if entry[kpt_id] != -1.0:
                kp[kpt_id, 0] = int(all_keypoints[int(entry[kpt_id]), 0])
                kp[kpt_id, 1] = int(all_keypoints[int(entry[kpt_id]), 1])
        poses.append(Pose(kp, entry[18]))
    return poses


def run_demo(net, image_provider, height_size, cpu, track, smooth, fused=False, draw=False):
    """Generator over frames: yields (img, current_poses).  No GUI (cv2.imshow/waitKey are out of scope)."""
    net = net.eval()
    stride, upsample_ratio = 8, 4
    previous_poses = []
    for img in image_provider:
        if fused:
            x, scale, pad = _prepare(net, img, height_size, stride, (0, 0, 0), (128, 128, 128), 1 / 256)
            pose_entries, all_keypoints, _ = net.engine.infer_poses(x, upsample_ratio, demo=True)[0]
        else:
            heatmaps, pafs, scale, pad = infer_fast(net, img, height_size, stride, upsample_ratio, cpu)
            total, by_type = 0, []
            for kpt_idx in range(Pose.num_kpts):      # the 19th map is background
                total += extract_keypoints(heatmaps[:, :, kpt_idx], by_type, total, engine=net.engine)
            pose_entries, all_keypoints = group_keypoints(by_type, pafs, demo=True, engine=net.engine)
        current_poses = poses_from_entries(pose_entries, all_keypoints, scale, pad, stride, upsample_ratio)
        if track:
            track_poses(previous_poses, current_poses, smooth=smooth)
            previous_poses = current_poses
        if draw:
            for pose in current_poses:
                pose.draw(img)
        yield img, current_poses
