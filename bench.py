#!/usr/bin/env python
"""bench.py — frames/sec end-to-end (net + key-point grouping) at 368 px on MI355X.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" is one pass of the hot path (network forward + peak extraction + PAF
grouping + result fetch to the host) over one batch of synthetic frames that are already resident in
HBM (normalised float32 NCHW).  Default workload = BASELINE.json configs[1]: batch 1, 368x656, one
refinement stage, fp32.  Frames shard across ranks with no data-path collective (weights are broadcast
once over RCCL at start-up), so scaling is weak: every rank processes `--batch` frames per step.

Timing protocol: W untimed warm-up steps, then BLOCKS of exactly K steps, each bracketed by a barrier +
device synchronisation on both sides and reduced with MAX over the ranks; blocks repeat until >= --min-time
seconds have been measured and `value` is the MEDIAN block (every block's figure is listed in `block_values`).

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline fields).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

_KEEP = []                   # keeps replica nets alive
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
MFMA_PEAK_TFLOPS = {"fp32": 157.3, "bf16": 2500.0}
METRIC = "frames/sec end-to-end (net+grouping) at 368px"


def layer_class(l):
    """Kernel family of a layer — each family is one kernel (template) name in a rocprof trace."""
    if l["kind"] == 0:
        return "stem"
    if l["kind"] == 3:
        return "fused_dw_pw"
    if l["kind"] == 1:
        return "depthwise"
    return "gemm_1x1" if l["ksize"] == 1 else "dense_3x3"


# kernel-name fragments of every family (rocprofv3 kernel_stats rows are matched with these)
KERNEL_NAMES = {"stem": ("stem_kernel",), "fused_dw_pw": ("dwpw_kernel", "dwpw_bf16_kernel", "dwpw_bf16_pp_kernel", "dwpw_pipe_kernel", "dwpw_tiled_kernel"), "depthwise": ("dw_kernel", "dw_tiled_kernel"),
                "gemm_1x1": ("gemm_ar_kernel", "gemm_wp_kernel", "gemm_kernel", "gemm_bf16_kernel", "gemm_bf16_ar_kernel", "heads_f32_kernel", "heads_bf16_kernel"),
                "dense_3x3": ("gemm_ar_kernel", "gemm_wp_kernel", "gemm_kernel", "gemm_bf16_kernel", "gemm_bf16_ar_kernel")}


def layer_work(layers, N, H, W, elt_bytes=4, covered=None):
    """Algorithmic FLOPs (2*MAC) and bytes (in + out + weights at the storage dtype) per kernel family, and per launch.
    ``covered``: {layer name: name of the layer whose LAUNCH also computes it} — a stage's second head conv inside the fused head
    kernel, a refinement block's `initial` 1x1 inside the preceding 3x3's epilogue (bf16, large M).  The tensor between the two
    is then never written or read, so it is not counted, and the covered layer's work belongs to the covering launch's family."""
    covered = covered or {}
    covers_something = set(covered.values())
    cls = {l["name"]: layer_class(l) for l in layers}
    h, w = H, W
    acc = {k: [0.0, 0.0] for k in ("stem", "depthwise", "fused_dw_pw", "gemm_1x1", "dense_3x3")}
    per_launch = {}
    for l in layers:
        hi, wi = h, w
        if l["stride"] == 2:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        m_out = N * h * w
        k = layer_class(l)
        if k == "stem":
            flops = 2.0 * m_out * 27 * 32
            byt = N * hi * wi * 3 * 4 + m_out * 32 * elt_bytes + 27 * 32 * 4       # f32 NCHW frame in, activations out
        elif k == "fused_dw_pw":   # depthwise 3x3 + pointwise 1x1 in one launch: both ops' FLOPs, block input + output + weights
            flops = 2.0 * m_out * l["macs_per_pixel"]
            byt = (N * hi * wi * l["cin"] + m_out * l["cout"] + 9 * l["cin"] + l["cin"] * l["cout"]) * elt_bytes
        elif k == "depthwise":
            flops = 2.0 * m_out * 9 * l["cin"]
            byt = (N * hi * wi * l["cin"] + m_out * l["cout"] + 9 * l["cin"]) * elt_bytes
        else:
            flops = 2.0 * m_out * l["macs_per_pixel"]        # merged heads: zero blocks are not counted
            a_in = 0 if l["name"] in covered else m_out * l["cin"]
            a_out = 0 if l["name"] in covers_something else m_out * l["cout"]
            byt = (a_in + a_out + l["macs_per_pixel"]) * elt_bytes      # weights = MACs per pixel (merged heads: the non-zero blocks)
            if l["res"] if "res" in l else False:
                byt += m_out * l["cout"] * elt_bytes
        owner = covered.get(l["name"], l["name"])
        acc[cls[owner]][0] += flops
        acc[cls[owner]][1] += byt
        e = per_launch.setdefault(owner, [cls[owner], 0.0, 0.0])
        e[1] += flops
        e[2] += byt
    return acc, [(n, v[0], v[1], v[2]) for n, v in per_launch.items()]


def rocprof_average_us(family, dtype):
    """Launch-weighted average duration of the family's kernels in the committed single-stream rocprofv3 summary
    (profiles/*/kernel_stats_b1_fp32_single_stream.csv) — shown beside the live HIP-event figure."""
    import csv
    import re
    root = os.path.join(ROOT, "profiles")
    cands = sorted(os.path.join(root, d, "kernel_stats_b1_fp32_single_stream.csv") for d in (os.listdir(root) if os.path.isdir(root) else []))
    cands = [c for c in cands if os.path.isfile(c)]
    if not cands or dtype != "fp32":
        return None, None
    tot_ns, calls, names = 0.0, 0, []
    for row in csv.DictReader(open(cands[-1])):
        nm = row["Name"]
        if not any(frag + "<" in nm for frag in KERNEL_NAMES[family]):
            continue
        if family in ("gemm_1x1", "dense_3x3") and "heads_" not in nm:
            m = re.search(r"<([^>]*)>", nm)
            last = m.group(1).split(",")[-1].strip() if m else ""
            if last != ("3" if family == "dense_3x3" else "1"):
                continue
        tot_ns += float(row["TotalDurationNs"])
        calls += int(row["Calls"])
        names.append(re.sub(r"^void\s+|lwp::|\(.*$", "", nm))
    if not calls:
        return None, None
    return tot_ns / calls / 1e3, {"source": os.path.relpath(cands[-1], ROOT), "kernels": names}


def cpu_baseline(sd, x, nref, budget_s=20.0):
    """The oracle (CPU restatement: stock torch.nn CPU kernels + NumPy post-processing) timed on this box's host cores on the
    same workload, demo.py --cpu semantics, one frame per iteration: the network at 1 thread and
    over a thread sweep up to all cores (a batch-1 forward does not scale to 128 threads: the best count is reported), the
    NumPy / Python post-processing stages (single-threaded by nature) once, median of the iterations."""
    import torch
    from oracle import net_ref, post_ref
    xt = torch.from_numpy(x)
    all_threads = torch.get_num_threads()

    def post(outs, stages=None):
        t0 = time.perf_counter()
        hu = post_ref.upsample_cubic(outs[-2][0].numpy().transpose(1, 2, 0), 4)
        pu = post_ref.upsample_cubic(outs[-1][0].numpy().transpose(1, 2, 0), 4)
        t1 = time.perf_counter()
        by_type, total = [], 0
        for k in range(18):
            total += post_ref.extract_keypoints(hu[:, :, k], by_type, total)
        t2 = time.perf_counter()
        post_ref.group_keypoints(by_type, pu, demo=True)
        t3 = time.perf_counter()
        if stages is not None:
            stages.append((t1 - t0, t2 - t1, t3 - t2))
        return total

    t_start = time.perf_counter()
    outs = net_ref.forward(sd, xt[0:1], nref)
    stages = []
    n_kpts = post(outs)
    for _ in range(3):
        post(outs, stages)
    up_ms, ex_ms, gr_ms = [float(np.median([s[i] for s in stages])) * 1e3 for i in range(3)]
    post_ms = up_ms + ex_ms + gr_ms
    sweep = []
    counts = sorted(set(t for t in (1, 4, 8, 16, 32, 64, all_threads) if t <= all_threads))
    share = max((budget_s - (time.perf_counter() - t_start)) / len(counts), 0.5)
    for t in counts:
        torch.set_num_threads(t)
        net_ref.forward(sd, xt[0:1], nref)
        ts, t0 = [], time.perf_counter()
        while len(ts) < 10 and (len(ts) < 2 or time.perf_counter() - t0 < share):
            f = len(ts) % xt.shape[0]
            t1 = time.perf_counter()
            net_ref.forward(sd, xt[f:f + 1], nref)
            ts.append(time.perf_counter() - t1)
        net_ms = float(np.median(ts)) * 1e3
        sweep.append({"threads": t, "net_ms": net_ms, "iters": len(ts), "frames_per_s": 1e3 / (net_ms + post_ms)})
    torch.set_num_threads(all_threads)
    best = max(sweep, key=lambda s: s["frames_per_s"])
    return {"value": best["frames_per_s"], "unit": "frames/s", "cores": best["threads"], "kind": "port",
            "sample": "one 368x656 synthetic frame per iteration (%d key-points): oracle net (torch CPU, median of %d forwards at the best "
                      "of %s threads) + NumPy post (median of 3); %.1f s of CPU work" % (n_kpts, best["iters"], counts, time.perf_counter() - t_start),
            "host_cores": os.cpu_count(), "torch_default_threads": all_threads,
            "stage_ms": {"net": best["net_ms"], "upsample_x4": up_ms, "extract_keypoints": ex_ms, "group_keypoints": gr_ms},
            "one_thread": {"frames_per_s": sweep[0]["frames_per_s"], "net_ms": sweep[0]["net_ms"], "post_ms": post_ms},
            "thread_sweep": sweep}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=1, help="frames per GPU per step")
    ap.add_argument("--height", type=int, default=368)
    ap.add_argument("--width", type=int, default=656)
    ap.add_argument("--nref", type=int, default=1)
    ap.add_argument("--dtype", default="fp32", choices=["fp32", "bf16"])
    ap.add_argument("--no-pipeline", action="store_true",
                    help="serialise the steps (submit, wait, fetch) instead of the default two-slot streaming pipeline, "
                         "in which the post-processing + host fetch of step k overlap the network of step k+1")
    ap.add_argument("--streams", type=int, default=3,
                    help="independent engine instances (own HIP streams and buffers) per GPU that take the steps round-robin; "
                         "every step is still one batch through the whole path.  At batch 1 one frame's kernels only half-fill "
                         "the chip, so frames in flight overlap (1 = single stream)")
    ap.add_argument("--min-time", type=float, default=1.0, help="repeat the timed block of --steps steps until this many seconds are measured")
    ap.add_argument("--max-blocks", type=int, default=400)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short measurements of the other BASELINE configs reported under other_configs")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--preroll", type=float, default=0.3, help="seconds of untimed passes before the warm-up steps (clock ramp)")
    return ap.parse_args()


def class_rooflines(eng, x, batch, height, width, dtype, dev_ms):
    """HIP events around EVERY launch of a pass (lwp_profile_launches), grouped into kernel families.  The events add a roughly
    constant gap per launch; the un-instrumented pipeline time (events around 20 whole passes) is the ground truth for the
    sum, so the per-launch overhead is (sum of per-launch times - pipeline time) / launches and it is removed from every launch.

    Roof of a family: bound time = max(algorithmic bytes / 8 TB/s, algorithmic FLOPs / MFMA peak of the dtype) over
    the family's launches; `bound` names the larger term, `achieved` / `peak` are in that roof's unit and frac = achieved / peak
    = bound time / measured time.  `frac_per_layer_roofs` takes the max per LAYER before summing (a family may mix both kinds)."""
    layers = eng.layers()
    fam = {l["name"]: layer_class(l) for l in layers}
    classes = {k: {"ms": 0.0, "launches": 0} for k in ("stem", "depthwise", "fused_dw_pw", "gemm_1x1", "dense_3x3", "post")}
    launched = set()
    for name, _, ms in eng.profile_launches(x, reps=10):
        c = classes[fam.get(name, "post")]
        c["ms"] += ms
        c["launches"] += 1
        launched.add(name)
    n_launch = sum(v["launches"] for v in classes.values())
    ev_overhead_ms = max(sum(v["ms"] for v in classes.values()) - dev_ms, 0.0) / max(n_launch, 1)
    for v in classes.values():
        v["ms_raw"] = v["ms"]
        v["ms"] = max(v["ms"] - ev_overhead_ms * v["launches"], 0.0)
    covered, last = {}, None                       # layers without a launch of their own ride in the previous launch
    for l in layers:
        if l["name"] in launched:
            last = l["name"]
        elif last is not None:
            covered[l["name"]] = last
    work, per_layer = layer_work(layers, batch, height, width, 4 if dtype == "fp32" else 2, covered)
    pk = MFMA_PEAK_TFLOPS[dtype]
    roofs = {}
    for k, (flops, byt) in work.items():
        ms, nl = classes[k]["ms"], max(classes[k]["launches"], 1)
        if ms <= 0 or flops <= 0:
            continue
        t_hbm, t_mfma = byt / (HBM_PEAK_GBS * 1e9), flops / (pk * 1e12)            # seconds
        t_mixed = sum(max(b_ / (HBM_PEAK_GBS * 1e9), f_ / (pk * 1e12)) for _, kk, f_, b_ in per_layer if kk == k)
        common = {"traffic": None, "avg_launch_us": ms * 1e3 / nl, "launches": nl, "alg_flops_per_launch": flops / nl,
                  "alg_bytes_per_launch": byt / nl, "flop_per_byte": flops / byt,
                  "hbm_bound_us_per_launch": t_hbm * 1e6 / nl, "mfma_bound_us_per_launch": t_mfma * 1e6 / nl,
                  "frac_per_layer_roofs": t_mixed / (ms * 1e-3),
                  "achieved_GBps": byt / (ms * 1e-3) / 1e9, "achieved_TFLOPs": flops / (ms * 1e-3) / 1e12}
        if k in ("depthwise", "stem") or t_hbm >= t_mfma:
            ach = byt / (ms * 1e-3) / 1e9
            roofs[k] = dict(common, bound="hbm", achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS)
        else:
            ach = flops / (ms * 1e-3) / 1e12
            roofs[k] = dict(common, bound="mfma", achieved=ach, peak=pk, unit="TFLOP/s", frac=ach / pk)
    return roofs, classes, ev_overhead_ms


def committed_traffic(name):
    """HBM-side traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of this
    very command line, summarised by tools/pmc_traffic.py with the gfx950 FETCH_SIZE x2 correction).  Counters cannot be read
    from inside the process, so the figure is the one measured when profiles/ was last refreshed."""
    pdir = os.path.join(ROOT, "profiles")
    cand = sorted(p_ for p_ in (os.path.join(pdir, d, name) for d in (os.listdir(pdir) if os.path.isdir(pdir) else [])) if os.path.isfile(p_))
    if not cand:
        return None, None
    try:
        return json.load(open(cand[-1]))["classes"], os.path.relpath(cand[-1], ROOT)
    except (OSError, ValueError, KeyError):
        return None, None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    import torch

    # LWP_BENCH_ENGINE_FACTORY="module:function": REHEARSAL of the N > 1 protocol on CPU-only machines (tests/): the factory
    # supplies stand-in engines, the process group is gloo, no device is touched and the line is marked "rehearsal".
    # LWP_BENCH_DEVICE / LWP_BENCH_BACKEND: rehearsal of the multi-rank path on a one-GPU box (all ranks on one card, gloo
    # instead of RCCL).  The driver's runs use one GPU per rank and "nccl" (= RCCL over xGMI).
    factory = os.environ.get("LWP_BENCH_ENGINE_FACTORY")
    rehearsal = factory is not None
    if os.environ.get("LWP_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["LWP_BENCH_DEVICE"])
    backend = "gloo" if rehearsal else os.environ.get("LWP_BENCH_BACKEND", "nccl")
    if not rehearsal:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(backend)
    red_dev = torch.device("cuda", local_rank) if (backend == "nccl" and not rehearsal) else torch.device("cpu")

    def sync():
        if not rehearsal:
            torch.cuda.synchronize()

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    if rehearsal:
        mod, fn = factory.split(":")
        engines, x, sd = getattr(importlib.import_module(mod), fn)(rank, world, args)
        eng = engines[0]
        x_np = None
    else:
        import lwpose_amd  # noqa: F401
        from lwpose_amd import dist as lwdist, synth, workload
        # weights: rank 0 builds + calibrates, then ONE RCCL broadcast of the packed blob (no per-step comms)
        net, sd = lwdist.build_replicated_net(args.nref, 1, local_rank, args.dtype, args.height, args.width, rank, world)
        eng = net.engine
        engines = [eng]
        if not args.no_pipeline and args.streams > 1:        # replicas of this rank's packed weights (device-to-device copy)
            from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
            blob = torch.empty(eng.weights_blob_bytes(), dtype=torch.uint8, device=torch.device("cuda", local_rank))
            eng.export_weights(blob)
            for _ in range(args.streams - 1):
                n2 = PoseEstimationWithMobileNet(args.nref, dtype=args.dtype)
                n2.eval().cuda(local_rank)
                n2.engine.import_weights(blob)
                engines.append(n2.engine)
                _KEEP.append(n2)
        # this rank's shard of the global batch: frames [rank*B, (rank+1)*B)
        frames = synth.make_frames(args.batch, args.height, args.width, seed0=rank * args.batch)
        x_np = workload.normalized_input(frames)
        x = torch.from_numpy(x_np).cuda(local_rank)

    def step():
        eng.infer_poses_async(x, 4, True)
        return eng.fetch_poses()

    def results_equal(a, b):
        return len(a) == len(b) and all(len(fa) == len(fb) and all(np.array_equal(u, v) for u, v in zip(fa, fb)) for fa, fb in zip(a, b))

    # every replica (engine streams of this rank; on ranks > 0 also engine 0, whose weights came over the broadcast) must give
    # engine 0's results on this rank's frames bit for bit, and every rank's packed weights must hash alike: a wrong replica
    # would still print a frames/s line
    if not rehearsal:
        ref_res = step()
        for e in engines[1:]:
            e.infer_poses_async(x, 4, True)
            if not results_equal(ref_res, e.fetch_poses()):
                sys.exit("bench: a replica engine's results differ from engine 0's on the same frames")
        replica_check = {"engines_checked": len(engines), "identical": True}
        if world > 1:
            blob_ = torch.empty(eng.weights_blob_bytes(), dtype=torch.uint8, device=torch.device("cuda", local_rank))
            eng.export_weights(blob_)
            b64 = blob_.view(torch.int64) if blob_.numel() % 8 == 0 else blob_.to(torch.int64)
            digest = torch.stack([b64.sum(), (b64 * torch.arange(1, b64.numel() + 1, device=b64.device)).sum()]).to(red_dev)
            lo, hi = digest.clone(), digest.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            if not torch.equal(lo, hi):
                sys.exit("bench: the ranks' packed weights differ after the broadcast")
            replica_check["ranks_with_equal_weight_digest"] = world
            del blob_, b64
    else:
        replica_check = None

    _TRACE = [] if os.environ.get("LWP_BENCH_TRACE") else None     # per-iteration host timestamps (diagnostics)

    def run_steps(k):
        """k complete passes (results of every pass fetched to the host)."""
        if args.no_pipeline:
            for _ in range(k):
                r = step()
            return r
        E = len(engines)
        pending = []
        r = None
        for i in range(k):
            if _TRACE is not None:
                _TRACE.append(time.perf_counter())
            e, slot = engines[i % E], (i // E) & 1
            if len(pending) >= 2 * E:                    # oldest step first: its slot is the one about to be reused
                pe, ps = pending.pop(0)
                r = pe.pipeline_fetch(ps)
            e.pipeline_submit(x, slot, 4, True)
            pending.append((e, slot))
        for pe, ps in pending:
            r = pe.pipeline_fetch(ps)
        return r

    # a full (generation-2) collection over torch's import graph costs ~40 ms on the submitting thread: take it now and
    # park the survivors, as a long-running streaming loop would, so it cannot land inside the timed region
    import gc
    gc.collect()
    gc.freeze()
    # pre-roll (not steps of the measurement): ~0.3 s of the same passes so that the first touch of every slot's
    # buffers and the GPU's clock ramp from idle are behind us whatever --warmup the caller passes
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.preroll:
        run_steps(2 * len(engines))
    res = run_steps(max(args.warmup, 1))

    def timed_block():
        """EXACTLY --steps steps between two (barrier + device sync) pairs; MAX over the ranks."""
        barrier()
        sync()
        t0 = time.perf_counter()
        r = run_steps(args.steps)
        sync()
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt, r

    blocks, total_t = [], 0.0
    while True:                                          # every rank sees the same (max-reduced) times: same number of blocks
        dt, res = timed_block()
        blocks.append(dt)
        total_t += dt
        if total_t >= args.min_time or len(blocks) >= args.max_blocks:
            break
    dt = sorted(blocks)[(len(blocks) - 1) // 2]          # median block (lower middle: an actual measurement)

    if _TRACE is not None:
        d = np.diff(np.array(_TRACE[-args.steps:])) * 1e3
        print("trace: first 24 iteration gaps (ms):", [round(float(v), 2) for v in d[:24]], "median", round(float(np.median(d)), 3),
              "slowest", [(int(i), round(float(d[i]), 2)) for i in np.argsort(d)[-5:]], file=sys.stderr)
    if rank == 0:
        frames_per_block = world * args.batch * args.steps
        in_flight = 1 if args.no_pipeline else len(engines)
        out = {
            "metric": METRIC,
            "value": frames_per_block / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "batch=%d per GPU, %dx%d frames, %d refinement stage(s), %s convs / f32+f64 post, "
                                   "resident normalised NCHW input -> key-points + pose entries on host; %d frame batch(es) in flight "
                                   "per GPU (%d engine stream(s), each step a complete pass)"
                                   % (args.batch, args.height, args.width, args.nref, args.dtype, in_flight, in_flight),
                       "global_batch": world * args.batch,
                       "parallelism": "dp%d (frames sharded, no data-path collective), %d stream(s) per GPU" % (world, len(engines))},
            "timed_blocks": len(blocks), "block_values": [frames_per_block / b for b in blocks],
            "block_spread": {"min": frames_per_block / max(blocks), "max": frames_per_block / min(blocks)},
            "pipelined": not args.no_pipeline, "streams": len(engines), "replica_check": replica_check,
            "poses_per_frame": float(np.mean([len(r[0]) for r in res])),
            "keypoints_per_frame": float(np.mean([len(r[1]) for r in res])),
        }
        if rehearsal:
            out["rehearsal"] = "stand-in engines from %s (no GPU touched): protocol check only, not a measurement" % factory
            out["roofline"] = None
            out["cpu_baseline"] = None
        else:
            measure_extras(out, args, eng, engines, net, sd, x, x_np, step, run_steps, local_rank, world)
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()              # rank 0 is still measuring its extra figures: leave together
        torch.distributed.destroy_process_group()


def measure_extras(out, args, eng, engines, net, sd, x, x_np, step, run_steps, local_rank, world):
    """Rank 0, after the timed region: device-only rates, per-class rooflines, CPU baseline, the other BASELINE configs."""
    import torch
    from lwpose_amd import synth, workload
    torch.cuda.synchronize()
    tl = time.perf_counter()
    for _ in range(20):
        step()
    latency_ms = (time.perf_counter() - tl) / 20 * 1e3       # serial step: submit, wait, fetch
    saved, engines[:] = list(engines), engines[:1]
    run_steps(10)
    torch.cuda.synchronize()
    tl = time.perf_counter()
    run_steps(100)
    torch.cuda.synchronize()
    single_stream_fps = 100 * args.batch / (time.perf_counter() - tl)
    engines[:] = saved
    dev_ms = eng.time_pipeline(x, 20, what=1) / 20.0
    net_ms = eng.time_pipeline(x, 20, what=0) / 20.0
    roofs, classes, ev_overhead_ms = class_rooflines(eng, x, args.batch, args.height, args.width, args.dtype, dev_ms)
    traffic_src = None
    std = (args.height, args.width, args.nref) == (368, 656, 1)
    if std and (args.batch, args.dtype) in ((1, "fp32"), (32, "fp32"), (32, "bf16")):
        pm, traffic_src = committed_traffic("pmc_traffic_b%d_%s.json" % (args.batch, args.dtype))
        if pm:
            for k in roofs:
                if k in pm:
                    roofs[k]["traffic"] = pm[k]["traffic_bytes_per_launch"]
    dominant = max(roofs, key=lambda k: classes[k]["ms"])
    rp_us, rp_src = rocprof_average_us(dominant, args.dtype) if (std and args.batch == 1) else (None, None)
    roof = dict(roofs[dominant], kernel=dominant, traffic_unit="bytes/launch (fabric-side, incl. Infinity-Cache hits)",
                traffic_source=traffic_src, rocprof_avg_launch_us=rp_us, rocprof=rp_src)
    out["roofline"] = roof
    out["roofline_classes"] = roofs
    out["latency_ms_serial_step"] = latency_ms
    out["single_stream_frames_per_s"] = single_stream_fps
    out["device_ms_per_step"] = {"pipeline": dev_ms, "network_only": net_ms, "post_only": max(dev_ms - net_ms, 0.0),
                                 "class_ms": {k: v["ms"] for k, v in classes.items()},
                                 "launches": {k: v["launches"] for k, v in classes.items()},
                                 "event_overhead_us_per_launch": ev_overhead_ms * 1e3}
    out["cpu_baseline"] = cpu_baseline(sd, x_np, args.nref, args.cpu_budget) if (world == 1 and not args.no_cpu_baseline) else None
    if world != 1 or args.no_extra_configs:
        return
    # the other BASELINE.json configs, same step definition, a few steps each (not the primary metric)
    other = {}
    out["other_configs"] = other

    def guarded(name, fn):
        try:
            other[name] = fn()
        except Exception as e:                    # a measurement of a secondary config must not take the bench line down
            other[name] = {"error": "%s: %s" % (type(e).__name__, e)}

    def replicas(e0, dt_, count):
        """`count` engines sharing e0's weights (device-to-device copy of the packed blob)."""
        from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
        out_ = [e0]
        if count > 1:
            blob = torch.empty(e0.weights_blob_bytes(), dtype=torch.uint8, device=torch.device("cuda", local_rank))
            e0.export_weights(blob)
            for _ in range(count - 1):
                n2 = PoseEstimationWithMobileNet(args.nref, dtype=dt_)
                n2.eval().cuda(local_rank)
                n2.engine.import_weights(blob)
                out_.append(n2.engine)
                _KEEP.append(n2)
        return out_

    def pipelined_steps(engs, xb, k):
        """k complete passes through the two-slot streaming pipeline of every engine (the headline's protocol)."""
        E, pending, r = len(engs), [], None
        for i in range(k):
            e, slot = engs[i % E], (i // E) & 1
            if len(pending) >= 2 * E:
                pe, ps = pending.pop(0)
                r = pe.pipeline_fetch(ps)
            e.pipeline_submit(xb, slot, 4, True)
            pending.append((e, slot))
        for pe, ps in pending:
            r = pe.pipeline_fetch(ps)
        return r

    def batched(b, dt_):
        def run():
            if dt_ == args.dtype:
                eng2 = eng
            else:
                net2, _ = workload.build_net(args.nref, 1, local_rank, dt_, args.height, args.width)
                eng2 = net2.engine
                _KEEP.append(net2)
            xb = torch.from_numpy(workload.normalized_input(synth.make_frames(b, args.height, args.width))).cuda(local_rank)
            # serial steps (submit, wait, fetch) ...
            for _ in range(2):
                eng2.infer_poses_async(xb, 4, True); rr = eng2.fetch_poses()
            ts = []
            for _ in range(3):                        # median of 3 blocks of 4 steps
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(4):
                    eng2.infer_poses_async(xb, 4, True); rr = eng2.fetch_poses()
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) / 4)
            t_serial = sorted(ts)[1]
            # ... and the headline's protocol: two engine streams, two result slots each (grouping + fetch of a batch overlap
            # the network of the next ones); every step is still a complete pass with its results on the host
            engs = replicas(eng2, dt_, int(os.environ.get("LWP_BENCH_BN_STREAMS", "2")))
            pipelined_steps(engs, xb, 4)
            # blocks long enough that the fill and the drain of the pipeline (four batches in flight) are a few per cent of a block:
            # 8-step blocks read 5-6 % below the sustained rate the headline loop measures on the same engines
            kb = 32 if b >= 32 else 64
            ts = []
            for _ in range(3):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                pipelined_steps(engs, xb, kb)
                torch.cuda.synchronize()
                ts.append((time.perf_counter() - t1) / kb)
            t2 = sorted(ts)[1]
            dms = eng2.time_pipeline(xb, 5, what=1) / 5.0
            ent = {"frames_per_s": b / t2, "ms_per_step": t2 * 1e3, "protocol": "%d engine stream(s) x 2 result slots (as the headline)" % len(engs),
                   "serial_frames_per_s": b / t_serial, "serial_ms_per_step": t_serial * 1e3, "device_ms_per_step": dms,
                   "poses_per_frame": float(np.mean([len(r[0]) for r in rr]))}
            if b == 32:
                rf, cl, _ = class_rooflines(eng2, xb, b, args.height, args.width, dt_, dms)
                pm, src = committed_traffic("pmc_traffic_b%d_%s.json" % (b, dt_)) if std else (None, None)
                if pm:
                    for k in rf:
                        if k in pm:
                            rf[k]["traffic"] = pm[k]["traffic_bytes_per_launch"]
                ent["roofline_classes"] = rf
                ent["traffic_source"] = src
            return ent
        return run
    for b, dt_ in ((8, "fp32"), (32, "fp32"), (32, "bf16")):
        if (b, dt_) != (args.batch, args.dtype):
            guarded("batch%d_%s" % (b, dt_), batched(b, dt_))

    def unfused_dw():
        """north_star's depthwise-conv HBM target: the stand-alone depthwise kernels (LWP_FUSE_DWPW=0) and the stem at batch 32."""
        os.environ["LWP_FUSE_DWPW"] = "0"
        try:
            net2, _ = workload.build_net(args.nref, 1, local_rank, "fp32", args.height, args.width, calibrate=False)
        finally:
            del os.environ["LWP_FUSE_DWPW"]
        xb = torch.from_numpy(workload.normalized_input(synth.make_frames(32, args.height, args.width))).cuda(local_rank)
        e2 = net2.engine
        e2.set_capacity(8192, 1024, 1 << 20, 4096)          # uncalibrated heads: noise-saturated maps (the post stage is not the subject here)
        dms = e2.time_pipeline(xb, 3, what=0) / 3.0
        rows = [(n, ms) for n, _, ms in e2.profile_launches(xb, reps=3)]
        layers = {l["name"]: l for l in e2.layers()}
        per_layer, tot_b, tot_ms = [], 0.0, 0.0
        h, w = args.height, args.width
        dims = {}
        for l in e2.layers():
            hi, wi = h, w
            if l["stride"] == 2:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
            dims[l["name"]] = (hi, wi, h, w)
        for n, ms in rows:
            l = layers.get(n)
            if l is None or l["kind"] != 1:
                continue
            hi, wi, ho, wo = dims[n]
            byt = (32 * hi * wi * l["cin"] + 32 * ho * wo * l["cout"] + 9 * l["cin"]) * 4
            per_layer.append({"layer": n, "us": ms * 1e3, "GB_per_s": byt / (ms * 1e-3) / 1e9})
            tot_b += byt
            tot_ms += ms
        stem_ms = [ms for n, ms in rows if n == "model.0"][0]
        stem_b = 32 * (args.height * args.width * 3 + dims["model.0"][2] * dims["model.0"][3] * 32) * 4
        return {"note": "per-launch HIP events include ~2-3 us of event overhead each", "network_ms": dms,
                "stem": {"us": stem_ms * 1e3, "GB_per_s": stem_b / (stem_ms * 1e-3) / 1e9, "frac_of_8TBs": stem_b / (stem_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "depthwise": {"GB_per_s": tot_b / (tot_ms * 1e-3) / 1e9, "frac_of_8TBs": tot_b / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                              "alg_MB_per_frame": tot_b / 32 / 1e6, "layers": per_layer}}
    guarded("batch32_fp32_unfused_depthwise", unfused_dw)

    def u8_input():
        """The true infer_fast() boundary (demo.py:59-66): the step starts from a uint8 frame in HOST memory: H2D of the frame,
        resize + normalize + pad on the device (lwp_preprocess_u8), then the same pass.  Serial steps, one stream."""
        fr = synth.make_frames(4, args.height, args.width, seed0=100)
        def one(i):
            xx, _, _ = eng.preprocess_u8(fr[i % 4], args.height, 8, hand_over=False)    # as demo._prepare: the tensor is internal to infer_fast
            eng.infer_poses_async(xx, 4, True)
            return eng.fetch_poses()
        for i in range(5):
            one(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(100):
            one(i)
        torch.cuda.synchronize()
        t2 = (time.perf_counter() - t1) / 100

        def piped(k):                             # the headline's protocol from the same boundary: len(engines) frames in flight
            E, pending, r = len(engines), [], None
            for i in range(k):
                e, slot = engines[i % E], (i // E) & 1
                if len(pending) >= 2 * E:
                    pe, ps = pending.pop(0)
                    r = pe.pipeline_fetch(ps)
                xx, _, _ = e.preprocess_u8(fr[i % 4], args.height, 8, hand_over=False)
                e.pipeline_submit(xx, slot, 4, True)
                pending.append((e, slot))
            for pe, ps in pending:
                r = pe.pipeline_fetch(ps)
            return r
        ent = {"frames_per_s": 1.0 / t2, "ms_per_step": t2 * 1e3, "note": "uint8 HxWx3 host frame -> H2D -> lwp_preprocess_u8 -> network + grouping -> host; serial, 1 stream"}
        if not args.no_pipeline and len(engines) > 1:
            piped(12)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            piped(300)
            torch.cuda.synchronize()
            t3 = (time.perf_counter() - t1) / 300
            ent["pipelined"] = {"frames_per_s": 1.0 / t3, "ms_per_step": t3 * 1e3, "note": "same boundary, %d frames in flight (the headline's engine streams)" % len(engines)}
        return ent
    if args.batch == 1 and args.dtype == "fp32":
        guarded("batch1_fp32_u8_input", u8_input)

    def config4():
        """BASELINE.json configs[3]: batch 32, 3 refinement stages, multi-scale [0.5, 1.0, 1.5] (val.py:81-134).  A step starts
        from 32 uint8 frames resident in HBM: per scale normalize + cubic resize + pad on the device, forward (368x368,
        368x656, 552x984), x8 up-sample + crop + resize + average, then extract / group at full resolution (demo=False)."""
        from lwpose_amd import val as lwval
        scales = [0.5, 1.0, 1.5]
        net3, _ = workload.build_net(3, 1, local_rank, "fp32", args.height, args.width, multiscale=scales)
        fr = torch.from_numpy(synth.make_frames(32, args.height, args.width, seed0=500)).cuda(local_rank)

        def ms_step():
            ah, ap_ = lwval.infer_batch(net3, fr, scales, args.height, 8)
            return lwval.poses_batch(net3, ah, ap_)
        ms_step()
        ts = []
        for _ in range(3):                    # median of three steps (a step is ~100 ms; one hiccup would dominate a mean)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            rr = ms_step()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t1)
        t2 = sorted(ts)[1]
        return {"frames_per_s": 32 / t2, "ms_per_step": t2 * 1e3, "gflop_per_frame": 212.9, "net_tflops": 32 * 212.9e9 / t2 / 1e12,
                "poses_per_frame": float(np.mean([len(r[0]) for r in rr])), "keypoints_per_frame": float(np.mean([len(r[1]) for r in rr])),
                "timed_region": "uint8 frames resident in HBM -> poses on host (image-side resize included)"}
    guarded("batch32_nref3_multiscale_fp32", config4)


if __name__ == "__main__":
    main()
