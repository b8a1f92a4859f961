"""Multi-GPU data parallelism for the inference path: one process per GPU, frames sharded by rank,
full weight replica per GPU.  The only collective is ONE broadcast of the packed weight blob at start-up
(torch.distributed, backend "nccl" = RCCL over xGMI); there is no steady-state communication
(the reference's only multi-GPU construct, nn.DataParallel at train.py:74, re-broadcasts every iteration).
"""
import torch


def shard_range(global_batch, rank, world):
    """Frames [lo, hi) of a global batch owned by ``rank`` (contiguous, remainder spread over the first ranks)."""
    q, r = divmod(global_batch, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def broadcast_weights(engine, rank, world, device, src=0):
    """Replicate rank ``src``'s packed weight blob to every rank's engine with one broadcast."""
    nbytes = engine.weights_blob_bytes()
    buf = torch.empty(nbytes, dtype=torch.uint8, device=device)
    if rank == src:
        engine.export_weights(buf)
    if world > 1:
        torch.distributed.broadcast(buf, src=src)
    if rank != src:
        engine.import_weights(buf)
    return nbytes


def build_replicated_net(nref, seed, local_rank, dtype, height, width, rank, world):
    """rank 0 builds + calibrates the synthetic net; the others receive its packed weights."""
    from . import workload
    if rank == 0:
        net, sd = workload.build_net(nref, seed, local_rank, dtype, height, width, calibrate=True)
    else:
        net, sd = workload.build_net(nref, seed, local_rank, dtype, height, width, calibrate=False)
    if world > 1:
        broadcast_weights(net.engine, rank, world, torch.device("cuda", local_rank))
    return net, sd


def gather_counts(values, world):
    """All-gather a small per-rank integer vector (pose counts / timings) — bookkeeping only."""
    t = torch.as_tensor(values, dtype=torch.int64)
    if world == 1:
        return [t]
    if torch.distributed.get_backend() == "nccl":
        t = t.cuda()
    out = [torch.empty_like(t) for _ in range(world)]
    torch.distributed.all_gather(out, t)
    return [o.cpu() for o in out]
