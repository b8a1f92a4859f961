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

Prints ONE JSON line on rank 0 (see DESIGN.md "Measurement" for the roofline / cpu_baseline fields).
"""
import argparse
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
KERNEL_NAMES = {"stem": ("stem_kernel",), "fused_dw_pw": ("dwpw_kernel", "dwpw_bf16_kernel"), "depthwise": ("dw_kernel",),
                "gemm_1x1": ("gemm_ar_kernel", "gemm_wp_kernel", "gemm_kernel", "gemm_bf16_kernel"),
                "dense_3x3": ("gemm_ar_kernel", "gemm_wp_kernel", "gemm_kernel", "gemm_bf16_kernel")}


def layer_work(layers, N, H, W, elt_bytes=4):
    """Algorithmic FLOPs (2*MAC) and bytes (in + out + weights at the storage dtype) per kernel family."""
    h, w = H, W
    acc = {k: [0.0, 0.0] for k in ("stem", "depthwise", "fused_dw_pw", "gemm_1x1", "dense_3x3")}
    for l in layers:
        hi, wi = h, w
        if l["stride"] == 2:
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        m_out = N * h * w
        k = layer_class(l)
        if k == "stem":
            flops = 2.0 * m_out * 27 * 32
            byt = (N * hi * wi * 3 + m_out * 32) * 4 + 27 * 32 * 4
        elif k == "fused_dw_pw":   # depthwise 3x3 + pointwise 1x1 in one launch: both ops' FLOPs, block input + output + weights
            flops = 2.0 * m_out * l["macs_per_pixel"]
            byt = (N * hi * wi * l["cin"] + m_out * l["cout"] + 9 * l["cin"] + l["cin"] * l["cout"]) * elt_bytes
        elif k == "depthwise":
            flops = 2.0 * m_out * 9 * l["cin"]
            byt = (N * hi * wi * l["cin"] + m_out * l["cout"] + 9 * l["cin"]) * elt_bytes
        else:
            flops = 2.0 * m_out * l["macs_per_pixel"]        # merged heads: zero blocks are not counted
            byt = (m_out * (l["cin"] + l["cout"]) + l["cin"] * l["cout"] * l["ksize"] ** 2) * elt_bytes
        acc[k][0] += flops
        acc[k][1] += byt
    return acc


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
        if family in ("gemm_1x1", "dense_3x3"):
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
    """The oracle (CPU restatement, stock torch.nn CPU kernels + NumPy post-processing) timed on this box's
    host cores on the same workload: demo.py --cpu semantics, one frame per iteration."""
    import torch
    from oracle import net_ref, post_ref

    def one(frame):
        outs = net_ref.forward(sd, frame, nref)
        hu = post_ref.upsample_cubic(outs[-2][0].numpy().transpose(1, 2, 0), 4)
        pu = post_ref.upsample_cubic(outs[-1][0].numpy().transpose(1, 2, 0), 4)
        by_type, total = [], 0
        for k in range(18):
            total += post_ref.extract_keypoints(hu[:, :, k], by_type, total)
        post_ref.group_keypoints(by_type, pu, demo=True)

    xt = torch.from_numpy(x)
    one(xt[0:1])
    t0 = time.perf_counter()
    n = 0
    while True:
        one(xt[n % xt.shape[0]:n % xt.shape[0] + 1])
        n += 1
        if time.perf_counter() - t0 > budget_s or n >= 200:
            break
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": "%d frames of the same 368x656 synthetic workload, %.1f s, oracle net (torch CPU) + NumPy post" % (n, dt)}


def main():
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
                         "the chip, so frames in flight overlap (default 3: 1.47k -> 1.96k -> 2.01k frames/s for 1 / 2 / 3 streams; 1 = single stream)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="skip the short batch-8/32 fp32 and batch-32 bf16 measurements reported under other_configs")
    ap.add_argument("--cpu-budget", type=float, default=15.0)
    ap.add_argument("--preroll", type=float, default=0.3, help="seconds of untimed passes before the warm-up steps (clock ramp)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            sys.exit("launch multi-GPU runs with torch.distributed.run (one process per GPU)")
    import torch
    import lwpose_amd  # noqa: F401
    from lwpose_amd import dist as lwdist, synth, workload

    # LWP_BENCH_DEVICE / LWP_BENCH_BACKEND: rehearsal of the multi-rank path on a one-GPU box (all ranks on one card,
    # gloo instead of RCCL); the driver's runs use one GPU per rank and "nccl" (= RCCL over xGMI)
    if os.environ.get("LWP_BENCH_DEVICE") is not None:
        local_rank = int(os.environ["LWP_BENCH_DEVICE"])
    backend = os.environ.get("LWP_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            torch.distributed.init_process_group(backend)

    # weights: rank 0 builds + calibrates, then ONE RCCL broadcast of the packed blob (no per-step comms)
    net, sd = lwdist.build_replicated_net(args.nref, 1, local_rank, args.dtype, args.height, args.width, rank, world)
    eng = net.engine
    engines = [eng]
    if not args.no_pipeline:
        from lwpose_amd.models.with_mobilenet import PoseEstimationWithMobileNet
        from lwpose_amd.modules.load_state import load_state
        if args.streams > 1:                             # replicas of this rank's packed weights (device-to-device copy)
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
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    if _TRACE is not None:
        d = np.diff(np.array(_TRACE[-args.steps:])) * 1e3
        print("trace: first 24 iteration gaps (ms):", [round(float(v), 2) for v in d[:24]], "median", round(float(np.median(d)), 3),
              "slowest", [(int(i), round(float(d[i]), 2)) for i in np.argsort(d)[-5:]], file=sys.stderr)
    if rank == 0:
        total_frames = world * args.batch * args.steps
        # device-only rates (HIP events on the engine's stream) and per-class launch times
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
        # HIP events around EVERY launch of a pass (lwp_profile_launches), grouped into kernel families.  The events add a
        # roughly constant gap per launch; the un-instrumented pipeline time (events around 20 whole passes) is the ground
        # truth for the sum, so the per-launch overhead is (sum of per-launch times - pipeline time) / launches and it
        # is removed from every launch.
        layers = eng.layers()
        fam = {l["name"]: layer_class(l) for l in layers}
        classes = {k: {"ms": 0.0, "launches": 0} for k in ("stem", "depthwise", "fused_dw_pw", "gemm_1x1", "dense_3x3", "post")}
        for name, _, ms in eng.profile_launches(x, reps=10):
            c = classes[fam.get(name, "post")]
            c["ms"] += ms
            c["launches"] += 1
        n_launch = sum(v["launches"] for v in classes.values())
        ev_overhead_ms = max(sum(v["ms"] for v in classes.values()) - dev_ms, 0.0) / max(n_launch, 1)
        for v in classes.values():
            v["ms_raw"] = v["ms"]
            v["ms"] = max(v["ms"] - ev_overhead_ms * v["launches"], 0.0)
        work = layer_work(layers, args.batch, args.height, args.width, 4 if args.dtype == "fp32" else 2)
        roofs = {}
        for k, (flops, byt) in work.items():
            ms, nl = classes[k]["ms"], max(classes[k]["launches"], 1)
            if ms <= 0 or flops <= 0:
                continue
            if k in ("depthwise", "stem"):
                ach = byt / (ms * 1e-3) / 1e9
                roofs[k] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                            "traffic": None, "avg_launch_us": ms * 1e3 / nl, "launches": nl, "alg_bytes_per_launch": byt / nl}
            else:
                ach = flops / (ms * 1e-3) / 1e12
                pk = MFMA_PEAK_TFLOPS[args.dtype]
                roofs[k] = {"bound": "mfma", "achieved": ach, "peak": pk, "unit": "TFLOP/s", "frac": ach / pk,
                            "traffic": None, "avg_launch_us": ms * 1e3 / nl, "launches": nl, "alg_flops_per_launch": flops / nl}
        # HBM-side traffic per launch from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate runs of
        # this very command line, summarised by tools/pmc_traffic.py with the gfx950 FETCH_SIZE x2 correction).  Counters
        # cannot be read from inside the process, so the figure is the one measured when profiles/ was last refreshed; it
        # is attached only when the workload is the one that was profiled.
        traffic_src = None
        if (args.batch, args.height, args.width, args.nref, args.dtype) == (1, 368, 656, 1, "fp32"):
            cand = sorted(p_ for p_ in (os.path.join(ROOT, "profiles", d, "pmc_traffic_b1_fp32.json") for d in
                                        (os.listdir(os.path.join(ROOT, "profiles")) if os.path.isdir(os.path.join(ROOT, "profiles")) else []))
                          if os.path.isfile(p_))
            if cand:
                try:
                    pm = json.load(open(cand[-1]))["classes"]
                    for k in roofs:
                        if k in pm:
                            roofs[k]["traffic"] = pm[k]["traffic_bytes_per_launch"]
                    traffic_src = os.path.relpath(cand[-1], ROOT)
                except (OSError, ValueError, KeyError):
                    traffic_src = None
        dominant = max(roofs, key=lambda k: classes[k]["ms"])
        rp_us, rp_src = rocprof_average_us(dominant, args.dtype) if (args.batch, args.height, args.width, args.nref) == (1, 368, 656, 1) else (None, None)
        roof = dict(roofs[dominant], kernel=dominant, traffic_unit="bytes/launch (fabric-side, incl. Infinity-Cache hits)",
                    traffic_source=traffic_src, rocprof_avg_launch_us=rp_us, rocprof=rp_src)
        byt_dom = work[dominant][1] / max(classes[dominant]["launches"], 1)
        roof["alg_bytes_per_launch"] = byt_dom
        out = {
            "metric": "frames/sec end-to-end (net+grouping) at 368px",
            "value": total_frames / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.dtype == "fp32" else "bf16", "data": "synthetic",
            "config": {"workload": "batch=%d per GPU, %dx%d frames, %d refinement stage(s), %s convs / f32+f64 post, "
                                   "resident normalised NCHW input -> key-points + pose entries on host"
                                   % (args.batch, args.height, args.width, args.nref, args.dtype),
                       "global_batch": world * args.batch,
                       "parallelism": "dp%d (frames sharded, no data-path collective), %d stream(s) per GPU" % (world, len(engines))},
            "roofline": roof,
            "roofline_classes": roofs,
            "latency_ms_serial_step": latency_ms, "single_stream_frames_per_s": single_stream_fps,
            "device_ms_per_step": {"pipeline": dev_ms, "network_only": net_ms, "post_only": max(dev_ms - net_ms, 0.0),
                                   "class_ms": {k: v["ms"] for k, v in classes.items()},
                                   "event_overhead_us_per_launch": ev_overhead_ms * 1e3},
            "pipelined": not args.no_pipeline, "streams": len(engines),
            "poses_per_frame": float(np.mean([len(r[0]) for r in res])),
            "keypoints_per_frame": float(np.mean([len(r[1]) for r in res])),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sd, x_np, args.nref, args.cpu_budget)
        else:
            out["cpu_baseline"] = None
        if world == 1 and not args.no_extra_configs:
            # the other BASELINE.json configs, same step definition, a few steps each (not the primary metric)
            other = {}
            for b, dt in ((8, "fp32"), (32, "fp32"), (32, "bf16")):
                if (b, dt) == (args.batch, args.dtype):
                    continue
                if dt == args.dtype:
                    net2, eng2 = net, eng
                else:
                    net2, _ = workload.build_net(args.nref, 1, local_rank, dt, args.height, args.width)
                    eng2 = net2.engine
                xb = torch.from_numpy(workload.normalized_input(synth.make_frames(b, args.height, args.width))).cuda(local_rank)
                for _ in range(2):
                    eng2.infer_poses_async(xb, 4, True); eng2.fetch_poses()
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(8):
                    eng2.infer_poses_async(xb, 4, True); eng2.fetch_poses()
                torch.cuda.synchronize()
                t2 = time.perf_counter() - t1
                other["batch%d_%s" % (b, dt)] = {"frames_per_s": 8 * b / t2, "ms_per_step": t2 / 8 * 1e3,
                                                  "device_ms_per_step": eng2.time_pipeline(xb, 5, what=1) / 5.0}
            # BASELINE.json configs[3]: batch 32, 3 refinement stages, multi-scale [0.5, 1.0, 1.5] (val.py:81-134): per step
            # three forwards (368x368, 368x656, 552x984 inputs, resident), x8 up-sample + crop + resize + average on the
            # device, then extract/group at full resolution with the demo=False rounding.  8 distinct frames tiled to 32
            # keep the host-side float resize of the inputs (outside the timed region, as in the reference) short.
            try:
                from lwpose_amd import val as lwval
                net3, _ = workload.build_net(3, 1, local_rank, "fp32", args.height, args.width)
                fr = list(synth.make_frames(8, args.height, args.width))
                ins = [(torch.from_numpy(np.tile(xs, (4, 1, 1, 1))).cuda(local_rank), pad)
                       for xs, pad in lwval.scaled_inputs(fr, [0.5, 1.0, 1.5], args.height, 8)]

                def ms_step():
                    ah, ap_ = lwval.accumulate_scales(net3, ins, args.height, args.width, 8)
                    return lwval.poses_batch(net3, ah, ap_)
                ms_step()
                ts = []
                for _ in range(3):                    # median of three steps (a step is ~130 ms; one hiccup would dominate a mean)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    rr = ms_step()
                    torch.cuda.synchronize()
                    ts.append(time.perf_counter() - t1)
                t2 = sorted(ts)[1]
                other["batch32_nref3_multiscale_fp32"] = {"frames_per_s": 32 / t2, "ms_per_step": t2 * 1e3, "gflop_per_frame": 212.9,
                                                          "net_tflops": 32 * 212.9e9 / t2 / 1e12,
                                                          "poses_per_frame": float(np.mean([len(r[0]) for r in rr]))}
            except Exception as e:                    # a measurement of a secondary config must not take the bench line down
                other["batch32_nref3_multiscale_fp32"] = {"error": "%s: %s" % (type(e).__name__, e)}
            out["other_configs"] = other
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()              # rank 0 is still measuring its extra figures: leave together
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
