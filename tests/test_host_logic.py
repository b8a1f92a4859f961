"""CPU: host-side logic — C-ABI exports, parameter table, pre-processing helpers, sharding, and the
world_size-2 weight-broadcast protocol over gloo (the N>1 path of bench.py without GPUs)."""
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

import lwpose_amd  # noqa: F401
from lwpose_amd import _lib, arch, dist as lwdist, synth
from lwpose_amd.val import normalize, pad_width
from oracle import preproc_ref

from conftest import GOLDEN, ROOT


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "lwpose.h")).read()
    declared = sorted(set(re.findall(r"\b(lwp_[a-z_0-9]+)\s*\(", hdr)))
    L = _lib.lib()
    for name in declared:
        assert hasattr(L, name), name
    assert sorted(declared) == sorted(_lib.EXPORTS)


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    rc = _lib.lib().lwp_create(0, 1, 128, 19, 38, 0, ctypes.byref(h))
    assert rc == _lib.LWP_ERR_NOGPU and not h.value
    with pytest.raises(RuntimeError):
        from lwpose_amd.runtime import Engine
        Engine(0)


@pytest.mark.parametrize("nref", [0, 1, 3])
def test_library_param_table_matches_python_and_reference(nref):
    spec = _lib.param_spec(nref)
    tab = arch.param_table(nref)
    assert [k for k, _, _ in spec] == [p.key for p in tab]
    assert [s for _, s, _ in spec] == [tuple(p.shape) for p in tab]
    if nref in (1, 3):
        keys = json.load(open(os.path.join(GOLDEN, "state_dict_keys_nref%d.json" % nref)))
        assert [k for k, _, _ in spec] == [k for k, _, _ in keys]
        assert [list(s) for _, s, _ in spec] == [s for _, s, _ in keys]


def test_bad_arguments_are_rejected_without_gpu():
    L = _lib.lib()
    assert L.lwp_param_count(-1, 128, 19, 38) == _lib.LWP_ERR_ARG
    h = ctypes.c_void_p()
    assert L.lwp_create(0, 1, 100, 19, 38, 0, ctypes.byref(h)) == _lib.LWP_ERR_ARG   # channels % 32
    assert b"multiple of 32" in L.lwp_last_error(None)
    assert L.lwp_destroy(None) == 0


def test_normalize_is_float64_and_exact():
    img = synth.make_frames(1, 16, 24)[0]
    out = normalize(img, (128, 128, 128), 1 / 256)
    assert out.dtype == np.float64                     # NumPy-2 promotion, as in the reference
    assert np.array_equal(out, (img.astype(np.float64) - 128) / 256)
    assert np.array_equal(out.astype(np.float32).astype(np.float64), out)


def test_pad_width_semantics():
    img = np.ones((368, 490, 3), np.float64)
    md = [368, max(490, 368)]
    out, pad = pad_width(img, 8, (0, 0, 0), md)
    assert md == [368, 496] and pad == [0, 3, 0, 3] and out.shape == (368, 496, 3)
    assert out[:, :3].sum() == 0 and out[:, -3:].sum() == 0 and out[:, 3:-3].min() == 1
    img = np.ones((184, 328, 3), np.float32)
    md = [368, max(328, 368)]
    out, pad = pad_width(img, 8, (0, 0, 0), md)      # val.infer at scale 0.5: padded up to base_height
    assert pad == [92, 20, 92, 20] and out.shape == (368, 368, 3)


def test_oracle_resize_cubic_u8_identity_range_and_float_cross_check():
    """The oracle's restatement of OpenCV's fixed-point uint8 cubic resize (demo.py:59; cv2 absent -> unpinned vs cv2):
    identity at scale 1, constants stay constant, and it stays within 1 grey level of the rounded float bicubic
    (torch, A = -0.75, same half-pixel mapping and border clamp)."""
    img = synth.make_frames(1, 40, 56)[0]
    assert np.array_equal(preproc_ref.resize_cubic_u8(img, 1.0, 1.0), img)
    const = np.full((20, 30, 3), 77, np.uint8)
    assert np.array_equal(preproc_ref.resize_cubic_u8(const, 1.7, 1.7), np.full((34, 51, 3), 77, np.uint8))
    smooth = torch.nn.functional.interpolate(torch.from_numpy(img.astype(np.float32)).permute(2, 0, 1)[None], scale_factor=4,
                                             mode="bilinear").clamp(0, 255).round()[0].permute(1, 2, 0).numpy().astype(np.uint8)
    for fx in (2.0, 1.5, 0.75):
        up = preproc_ref.resize_cubic_u8(smooth, fx, fx)
        t = torch.from_numpy(smooth.astype(np.float32)).permute(2, 0, 1)[None]
        ref = torch.nn.functional.interpolate(t, size=up.shape[:2], scale_factor=None, mode="bicubic", align_corners=False)
        if fx == 2.0:                                # torch derives the same scale from the sizes only when they divide exactly
            ref = ref.clamp(0, 255)[0].permute(1, 2, 0).numpy()
            assert up.shape == (320, 448, 3) and up.dtype == np.uint8
            assert np.abs(up.astype(np.float32) - ref).max() <= 1.0 + 1e-3
        assert up.dtype == np.uint8 and up.shape[2] == 3


@pytest.mark.parametrize("H,W,net_h,stride", [(368, 656, 368, 8), (480, 640, 368, 8), (200, 300, 368, 8), (721, 1283, 368, 8),
                                               (1080, 1920, 256, 8), (333, 111, 368, 8), (64, 96, 64, 16)])
def test_preprocess_dims_match_the_oracle(H, W, net_h, stride):
    """lwp_preprocess_dims (pure host arithmetic of the C-ABI library) against the oracle's demo.py:55-62 restatement."""
    from lwpose_amd.runtime import Engine
    img = np.zeros((H, W, 3), np.uint8)
    x, scale, pad = preproc_ref.prepare_frame(img, net_h, stride)
    dh, dw, oh, ow, pad2, scale2 = Engine.preprocess_dims(H, W, net_h, stride)
    assert (oh, ow) == x.shape[2:] and pad2 == pad and scale2 == scale
    assert (dh, dw) == (int(round(H * scale)), int(round(W * scale)))


def test_convert_to_coco_format_matches_reference_outputs():
    """val.convert_to_coco_format vs the reference's own function run on the reference's grouping results
    (tests/golden/coco_format.json, generated by oracle/make_golden.py gen_coco_format)."""
    from lwpose_amd.val import convert_to_coco_format
    ref = json.load(open(os.path.join(GOLDEN, "coco_format.json")))
    assert sum(len(v["scores"]) for v in ref.values()) >= 20
    for name, want in ref.items():
        g = np.load(os.path.join(GOLDEN, "post_%s.npz" % name))
        kps, scores = convert_to_coco_format(g["val_entries"], g["val_allk"])
        assert [[float(v) for v in k] for k in kps] == want["keypoints"]
        assert [float(v) for v in scores] == want["scores"]


def test_one_euro_filter_matches_reference_sequence():
    from lwpose_amd.modules.one_euro_filter import OneEuroFilter
    ref = json.load(open(os.path.join(GOLDEN, "one_euro.json")))
    f = OneEuroFilter(freq=15, beta=0.1)
    assert [float(f(x)) for x in ref["x"]] == ref["y"]


def test_pose_bbox_and_tracking_ids():
    from lwpose_amd.modules.pose import Pose, track_poses
    kp = -np.ones((18, 2), np.int32)
    kp[0] = (10, 20); kp[1] = (14, 28); kp[5] = (7, 25)
    p = Pose(kp, 3.5)
    assert p.bbox == (7, 20, 8, 9)
    Pose.last_id = -1
    a = [Pose(kp.copy(), 3.0)]
    track_poses([], a)
    kp2 = kp.copy()
    b = [Pose(kp2, 2.0)]
    track_poses(a, b, smooth=True)
    assert a[0].id == 0 and b[0].id == 0


def test_shard_range_partitions_the_batch():
    for gb in (1, 7, 32, 256):
        for world in (1, 2, 3, 8):
            spans = [lwdist.shard_range(gb, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == gb
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, numpy as np
import lwpose_amd
from lwpose_amd import dist as lwdist
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
torch.distributed.init_process_group("gloo")
class FakeEngine:                       # stands in for runtime.Engine: the blob protocol without a GPU
    def __init__(self, fill): self.blob = torch.full((4096,), fill, dtype=torch.uint8)
    def weights_blob_bytes(self): return self.blob.numel()
    def export_weights(self, t): t.copy_(self.blob)
    def import_weights(self, t): self.blob = t.clone()
eng = FakeEngine(7 if rank == 0 else 0)
if rank == 0: eng.blob[:16] = torch.arange(16, dtype=torch.uint8)
n = lwdist.broadcast_weights(eng, rank, world, torch.device("cpu"))
exp = torch.full((4096,), 7, dtype=torch.uint8); exp[:16] = torch.arange(16, dtype=torch.uint8)
assert n == 4096 and torch.equal(eng.blob, exp), rank
lo, hi = lwdist.shard_range(9, rank, world)
got = lwdist.gather_counts([rank, hi - lo], world)
assert [int(g[1]) for g in got] == [5, 4] and [int(g[0]) for g in got] == [0, 1]
# max-over-ranks timing reduction used by bench.py
t = torch.tensor([1.0 + rank], dtype=torch.float64)
torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
assert float(t) == 2.0
torch.distributed.barrier()
torch.distributed.destroy_process_group()
open(os.path.join(sys.argv[2], "ok_%d" % rank), "w").write("ok")
'''


def test_world_size_2_broadcast_and_sharding_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = 29500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(script), ROOT, str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "ok_0").exists() and (tmp_path / "ok_1").exists()


def test_bf16_rejects_unsupported_channel_widths():
    """The bf16 graph exists for num_channels in {64, 128, 256, 512} only: anything else is an argument error
    at lwp_create, before any device is touched — never f32 kernels on bf16-sized buffers."""
    L = _lib.lib()
    for C in (96, 160, 192, 32):
        h = ctypes.c_void_p()
        assert L.lwp_create(0, 1, C, 19, 38, _lib.BF16, ctypes.byref(h)) == _lib.LWP_ERR_ARG and not h.value
        assert b"bf16 path supports" in L.lwp_last_error(None)
    from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
    net = PoseEstimationWithMobileNet(1, num_channels=96, dtype="bf16")
    with pytest.raises(ValueError):
        net.cuda()


def test_scale_dims_match_the_oracle_pad_width():
    """lwp_scale_dims (host arithmetic of val.py:89-91) against the oracle's pad_width on the scaled size."""
    from lwpose_amd.runtime import Engine
    for H, W, ratio, base, stride in ((368, 656, 0.5, 368, 8), (368, 656, 1.5, 368, 8), (92, 120, 3.0, 184, 8), (75, 333, 2.45, 368, 16),
                                      (5, 7, 0.5, 368, 8)):
        dh, dw = int(round(H * ratio)), int(round(W * ratio))
        _, pad = preproc_ref.pad_width(np.zeros((dh, dw, 3)), stride, (0, 0, 0), [base, max(dw, base)])
        got = Engine.scale_dims(H, W, ratio, base, stride)
        assert got[:2] == (dh, dw) and got[4] == pad and got[2:4] == (dh + pad[0] + pad[2], dw + pad[1] + pad[3])


def test_checkpoint_round_trip_host_side(tmp_path, capsys):
    """{'state_dict': ...} written by torch.save is read back with the weights-only loader (nothing in the file is executed)
    and poured in through load_state: every tensor arrives, a missing key keeps the net's own value and prints the
    reference's warning (modules/load_state.py:4-15)."""
    from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
    from lwpose_amd.modules.load_state import load_state
    sd = synth.make_state_dict(1, seed=5)
    del sd["cpm.conv.0.bias"]
    path = str(tmp_path / "ck.pth")
    torch.save({"state_dict": sd, "iter": 370000}, path)
    ck = torch.load(path, map_location="cpu", weights_only=True)
    net = PoseEstimationWithMobileNet(1)
    own = net.state_dict()["cpm.conv.0.bias"].clone()
    load_state(net, ck)
    assert capsys.readouterr().out == "[WARNING] Not found pre-trained parameters for cpm.conv.0.bias\n"
    after = net.state_dict()
    assert torch.equal(after["cpm.conv.0.bias"], own)
    assert all(torch.equal(after[k], v) for k, v in sd.items())


# ------------------------------------------------------------------------------------------ tail of the path: oracle pins + product
def test_tail_oracle_hand_cases():
    """oracle/tail_ref.py against values derived by hand from demo.py:101-114 and pose.py:21-45 (cv2 is absent, so the
    reference's own functions cannot be imported: 'parity unpinned' beyond these cases)."""
    from oracle import tail_ref
    allk = np.array([[100.0, 40.0, 0.9, 0], [37.0, 81.0, 0.8, 1], [2.0, 1.0, 0.7, 2]])
    ent = -np.ones((2, 20)); ent[0, 0] = 0; ent[0, 1] = 1; ent[0, 18] = 3.25; ent[0, 19] = 2
    ent[1, 5] = 2; ent[1, 18] = 1.5; ent[1, 19] = 1
    poses = tail_ref.poses_from_entries(ent, allk.copy(), 0.8, [4, 6, 4, 6], 8, 4)
    # x = (x * 8 / 4 - pad[1]) / scale ; y = (y * 8 / 4 - pad[0]) / scale ; int() truncates toward zero
    assert poses[0].keypoints[0].tolist() == [int((200 - 6) / 0.8), int((80 - 4) / 0.8)] == [242, 95]
    assert poses[0].keypoints[1].tolist() == [int((74 - 6) / 0.8), int((162 - 4) / 0.8)] == [85, 197]
    assert poses[1].keypoints[5].tolist() == [int((4 - 6) / 0.8), int((2 - 4) / 0.8)] == [-2, -2]       # truncation, not floor
    assert (poses[0].keypoints[2:] == -1).all() and poses[0].confidence == 3.25
    assert poses[0].bbox == (85, 95, 242 - 85 + 1, 197 - 95 + 1) and poses[1].bbox == (-2, -2, 1, 1)
    tail_ref.RefPose.last_id = -1
    poses[0].update_id(); poses[1].update_id(); poses[0].update_id(7)
    assert (poses[0].id, poses[1].id, tail_ref.RefPose.last_id) == (7, 1, 1)
    f = tail_ref.RefOneEuro(freq=15, beta=0.1)
    ref = json.load(open(os.path.join(GOLDEN, "one_euro.json")))
    assert [float(f(x)) for x in ref["x"]] == ref["y"]                   # the filter restatement IS pinned to the reference


def _replay_tracking_case(case, make_pose, track, last_id_get, last_id_set):
    last_id_set(-1)
    prev = []
    for t, fr in enumerate(case["frames"]):
        poses = [make_pose(np.array(kp, dtype=np.int32), c) for kp, c in zip(fr["in_keypoints"], fr["in_confidence"])]
        if not (t == 0 and not case["first_frame_ids"]):
            track(prev, poses, threshold=case["threshold"], smooth=case["smooth"])
        assert [p.id for p in poses] == fr["ids"], (case["name"], t)
        for p, kp, bb in zip(poses, fr["keypoints"], fr["bbox"]):
            assert np.array_equal(p.keypoints, np.array(kp, dtype=np.int32)), (case["name"], t)
            assert [int(v) for v in p.bbox] == bb, (case["name"], t)
        assert last_id_get() == fr["last_id"], (case["name"], t)
        prev = poses


def test_track_poses_and_similarity_match_the_reference_goldens():
    """tests/golden/tracking.json holds the outputs of the REFERENCE's own get_similarity / track_poses / Pose.update_id bodies
    (modules/pose.py:21-27,41-45,65-118, captured by oracle/make_golden.py:gen_tracking; only cv2.boundingRect is restated):
    19 sequences x smooth x thresholds (incl. <= 0), previous poses without ids, poses similar to nobody, an all-missing pose.
    BOTH the product (modules/pose.py of this package) and the oracle restatement (oracle/tail_ref.py) must reproduce them."""
    from lwpose_amd.modules import pose as prod
    from oracle import tail_ref
    g = json.load(open(os.path.join(GOLDEN, "tracking.json")))
    assert len(g["cases"]) >= 19
    for case in g["cases"]:
        _replay_tracking_case(case, prod.Pose, prod.track_poses, lambda: prod.Pose.last_id, lambda v: setattr(prod.Pose, "last_id", v))
        _replay_tracking_case(case, tail_ref.RefPose, tail_ref.track_poses, lambda: tail_ref.RefPose.last_id,
                              lambda v: setattr(tail_ref.RefPose, "last_id", v))
    for sm in g["similarity"]:
        for mk, fn in ((prod.Pose, prod.get_similarity), (tail_ref.RefPose, tail_ref.get_similarity)):
            a = [mk(np.array(kp, dtype=np.int32), 1.0) for kp in sm["a"]]
            b = [mk(np.array(kp, dtype=np.int32), 1.0) for kp in sm["b"]]
            assert [[int(fn(p, q, sm["threshold"])) for q in b] for p in a] == sm["counts"]
            assert [int(fn(p, p, sm["threshold"])) for p in a] == sm["self"]


def _people_sequence(seed, n_frames=5, n_people=4, drop=0.15):
    """Seeded walk of ``n_people`` skeletons: per frame a list of ((18,2) int32 key-points, confidence)."""
    rng = np.random.RandomState(seed)
    base = rng.randint(40, 400, size=(n_people, 1, 2)) + rng.randint(-30, 31, size=(n_people, 18, 2))
    vel = rng.randint(-6, 7, size=(n_people, 1, 2))
    frames = []
    for t in range(n_frames):
        cur = []
        order = rng.permutation(n_people)
        for p in order[: n_people - (1 if t == 2 else 0)]:          # one person vanishes in frame 2 and comes back
            kp = (base[p] + vel[p] * t + rng.randint(-2, 3, size=(18, 2))).astype(np.int32)
            kp[rng.rand(18) < drop] = -1
            cur.append((kp, float(rng.rand() * 10)))
        if t == 3:
            cur.append((rng.randint(500, 600, size=(18, 2)).astype(np.int32), 0.5))   # a newcomer far away
        frames.append(cur)
    return frames


@pytest.mark.parametrize("smooth", [False, True])
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_track_poses_matches_oracle_on_sequences(seed, smooth):
    """modules.pose.track_poses / Pose vs the oracle restatement of pose.py:65-118 on seeded 5-frame, 4-person sequences:
    ids, greedy mask order (most confident first, first maximum wins), >= threshold rule, filter hand-over only where the
    previous pose has the key-point, smoothed key-points and refreshed boxes."""
    from lwpose_amd.modules.pose import Pose, track_poses
    from oracle import tail_ref
    Pose.last_id = -1
    tail_ref.RefPose.last_id = -1
    prev_a, prev_b = [], []
    for cur in _people_sequence(seed):
        a = [Pose(kp.copy(), c) for kp, c in cur]
        b = [tail_ref.RefPose(kp.copy(), c) for kp, c in cur]
        track_poses(prev_a, a, smooth=smooth)
        tail_ref.track_poses(prev_b, b, smooth=smooth)
        assert [p.id for p in a] == [p.id for p in b]
        for p, q in zip(a, b):
            assert np.array_equal(p.keypoints, q.keypoints) and tuple(p.bbox) == tuple(q.bbox)
            if smooth:
                for k in range(18):
                    assert (p.filters[k][0].x_previous, p.filters[k][0].dx) == (q.filters[k][0].raw_prev, q.filters[k][0].dx)
        prev_a, prev_b = a, b
    assert Pose.last_id == tail_ref.RefPose.last_id >= 4
    ids = [p.id for p in prev_a]
    assert len(set(ids)) == len(ids)


def test_track_poses_corner_cases_match_oracle():
    """threshold <= 0 with nothing similar (NumPy's mask[None] = 0 clears every flag), previous poses without ids, and
    get_similarity on identical / disjoint poses."""
    from lwpose_amd.modules.pose import Pose, get_similarity, track_poses
    from oracle import tail_ref
    fr = _people_sequence(9, n_frames=2)
    for threshold, give_ids in ((0, True), (3, False), (19, True), (1, True)):
        Pose.last_id = tail_ref.RefPose.last_id = -1
        pa = [Pose(kp.copy(), c) for kp, c in fr[0]]
        pb = [tail_ref.RefPose(kp.copy(), c) for kp, c in fr[0]]
        if give_ids:
            track_poses([], pa); tail_ref.track_poses([], pb)
        far = [(kp + 5000, c) for kp, c in fr[1][:2]] + fr[1][2:]          # two poses similar to nobody
        ca = [Pose(kp.copy(), c) for kp, c in far]
        cb = [tail_ref.RefPose(kp.copy(), c) for kp, c in far]
        track_poses(pa, ca, threshold=threshold, smooth=True)
        tail_ref.track_poses(pb, cb, threshold=threshold, smooth=True)
        assert [p.id for p in ca] == [p.id for p in cb], (threshold, give_ids)
        assert all(np.array_equal(p.keypoints, q.keypoints) for p, q in zip(ca, cb))
    a, b = Pose(fr[0][0][0].copy(), 1.0), tail_ref.RefPose(fr[0][0][0].copy(), 1.0)
    assert get_similarity(a, a) == tail_ref.get_similarity(b, b) == int((fr[0][0][0][:, 0] != -1).sum())


def test_bench_multi_rank_protocol_over_gloo(tmp_path):
    """The REAL bench.py N > 1 code path (init_process_group, one weight broadcast, per-rank shards, barrier-bracketed
    timed blocks of exactly K steps, MAX over ranks, one JSON line on rank 0) with world_size 2 over gloo; the engines are
    sleeping stand-ins (tests/stub_engine.py), rank 1 twice as slow as rank 0."""
    port = 31500 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "1", "--batch", "3",
           "--min-time", "0.1", "--preroll", "0.01"]
    env = dict(os.environ, OMP_NUM_THREADS="1", LWP_BENCH_ENGINE_FACTORY="tests.stub_engine:make")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                       # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["global_batch"] == 6 and out["unit"] == "frames/s" and out["higher_is_better"] is True
    assert "rehearsal" in out and out["roofline"] is None and out["cpu_baseline"] is None
    from tests.stub_engine import STEP_SECONDS
    # the slow rank (2 x STEP_SECONDS per step) sets the time: max over ranks, not rank 0's own clock
    assert out["ms_per_step"] >= 2 * STEP_SECONDS * 1e3 * 0.98
    assert abs(out["value"] - 6 * 6 / (out["ms_per_step"] * 6 / 1e3)) < 1e-6 * out["value"]
    assert out["timed_blocks"] == len(out["block_values"]) >= 2
    assert out["block_spread"]["min"] <= out["value"] <= out["block_spread"]["max"]


def test_pose_draw_marks_joints_and_limbs_and_stays_inside_the_image():
    """modules/pose.py:47-62 (`Pose.draw`: circles on the found key-points, a line per limb whose two ends were found).  The
    rasteriser is the product's own (cv2 is absent): joints and limbs must be painted in `Pose.color`, missing key-points
    (-1) and limbs with a missing end must not be, and points at / beyond the border must not raise or wrap around."""
    from lwpose_amd.modules.pose import BODY_PARTS_KPT_IDS, Pose
    kp = np.full((Pose.num_kpts, 2), -1, np.int32)
    a, b = BODY_PARTS_KPT_IDS[0]
    kp[a] = (10, 12)
    kp[b] = (30, 12)                                   # a horizontal limb
    c, d = BODY_PARTS_KPT_IDS[2]
    if c not in (a, b):
        kp[c] = (0, 0)                                 # a joint on the corner; its partner stays missing
    img = np.zeros((40, 48, 3), np.uint8)
    Pose(kp, 1.0).draw(img)
    col = np.array(Pose.color, np.uint8)
    assert (img[12, 10] == col).all() and (img[12, 30] == col).all()          # the joints
    assert (img[12, 11:30] == col).all()                                      # every pixel of the limb between them
    assert not img[30:, :].any() and not img[:, 40:].any()                    # nothing far from the pose
    painted = np.argwhere(img.any(axis=2))
    assert painted[:, 0].max() <= 12 + 3 and painted[:, 1].max() <= 30 + 3
    if c not in (a, b):
        assert (img[0, 0] == col).all()
    far = np.full((Pose.num_kpts, 2), -1, np.int32)
    far[a] = (47, 39)
    far[b] = (60, 50)                                  # outside the image: clipped, no exception
    Pose(far, 1.0).draw(img)
    assert (img[39, 47] == col).all()
