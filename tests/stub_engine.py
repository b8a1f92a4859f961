"""Stand-in engines for the CPU REHEARSAL of bench.py's multi-rank protocol (LWP_BENCH_ENGINE_FACTORY=tests.stub_engine:make).

Not a compute path and not part of the product: the stub sleeps instead of launching kernels, so that
tests/test_host_logic.py can drive the real bench.py N > 1 code (process group, one weight broadcast, sharding by rank,
barrier-bracketed timed blocks, MAX over ranks, the JSON line) with world_size 2 over gloo on a machine without GPUs."""
import time

import numpy as np
import torch

STEP_SECONDS = 0.002          # rank r sleeps (1 + r) x this per step: the slowest rank must set the reported time


class StubEngine(object):
    def __init__(self, rank, batch):
        self.rank, self.batch = rank, batch
        self.blob = torch.full((1024,), 42 if rank == 0 else 0, dtype=torch.uint8)
        self.submitted = {}
        self.steps = 0

    # --- weight replication protocol (dist.broadcast_weights)
    def weights_blob_bytes(self):
        return self.blob.numel()

    def export_weights(self, t):
        t.copy_(self.blob)

    def import_weights(self, t):
        self.blob = t.clone()

    # --- the calls bench.py's step loop makes
    def _result(self):
        return [(np.zeros((2, 20)), np.zeros((5, 4)), np.zeros(18, np.int32)) for _ in range(self.batch)]

    def pipeline_submit(self, x, slot, ratio, demo):
        assert slot not in self.submitted, "slot reused before it was fetched"
        assert tuple(x.shape[:2]) == (self.batch, 3)
        self.submitted[slot] = True

    def pipeline_fetch(self, slot):
        assert self.submitted.pop(slot)
        time.sleep(STEP_SECONDS * (1 + self.rank))
        self.steps += 1
        return self._result()

    def infer_poses_async(self, x, ratio, demo):
        self._pending = True

    def fetch_poses(self):
        time.sleep(STEP_SECONDS * (1 + self.rank))
        self.steps += 1
        return self._result()


def make(rank, world, args):
    from lwpose_amd import dist as lwdist
    eng = StubEngine(rank, args.batch)
    lwdist.broadcast_weights(eng, rank, world, torch.device("cpu"))
    assert int(eng.blob[0]) == 42, "weight blob did not arrive"
    lo, hi = lwdist.shard_range(world * args.batch, rank, world)
    assert (lo, hi) == (rank * args.batch, (rank + 1) * args.batch)
    x = torch.zeros((args.batch, 3, 8, 8), dtype=torch.float32)
    return [eng], x, None
