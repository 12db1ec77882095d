"""Multi-GPU plumbing: one process per GPU, population sharded by rows.

Takes the place of the reference's MPI backend (stochopy/optimize/_common.py:45-72: every
rank runs the optimiser, rank 0's candidates are broadcast, fitness is summed with Allreduce).
Here each GPU owns ``popsize / world`` rows for the whole run; per generation the only traffic is
ONE all-gather of an (n+2)-double record per rank -- [best f, global row, best row] -- over RCCL
(``torch.distributed`` backend "nccl" on ROCm = RCCL over xGMI), after which every rank finalises
the same global best locally.  The population never moves.

Semantics with ``workers > 1`` (documented deviation): DE donors are drawn
from the rank's own shard (an island model with a shared global best); PSO is exact.  Random draws
must be ``rng="philox"`` (counter-based, keyed by the GLOBAL row, so they do not depend on the
sharding).

With the "gloo" backend the record is staged through host memory -- used by the tests (two ranks on
one GPU, or CPU-only plumbing checks); production uses "nccl".
"""
import numpy as np


def require_world(workers):
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        raise RuntimeError(
            f"workers={workers}: launch one process per GPU (torchrun / torch.distributed.run) and call "
            "torch.distributed.init_process_group first (see bench.py)")
    if dist.get_world_size() != workers:
        raise RuntimeError(f"workers={workers} but the process group has {dist.get_world_size()} ranks")
    return World(dist)


def shard_bounds(popsize, world, rank):
    """Rows [row0, row0 + count) of rank `rank`: equal shards (popsize must divide evenly)."""
    if popsize % world != 0:
        raise ValueError(f"popsize={popsize} must be a multiple of workers={world}")
    count = popsize // world
    return rank * count, count


class World:
    """The process group as the generation loop sees it."""

    def __init__(self, dist, group=None):
        self.dist = dist
        self.group = group
        self.size = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.backend = dist.get_backend(group)

    def shard(self, popsize):
        return shard_bounds(popsize, self.size, self.rank)

    def all_gather_records(self, record, out):
        """out[(world, n+2)] <- every rank's record[(n+2,)].  Device tensors; asynchronous with "nccl"."""
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(out.view(-1), record, group=self.group)
            return
        # host-staged exchange (gloo): tests only
        import torch

        host = record.detach().cpu()
        gathered = [torch.empty_like(host) for _ in range(self.size)]
        self.dist.all_gather(gathered, host, group=self.group)
        out.copy_(torch.stack(gathered).to(out.device))

    def max_over_ranks(self, value):
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


def best_of_records(records):
    """Host restatement of the device rule (sx_gather_finalize): first minimum by (f, global row).
    records: (world, n+2) array-like.  Returns (winner rank, f, global row)."""
    rec = np.asarray(records)
    order = np.lexsort((rec[:, 1], rec[:, 0]))
    w = int(order[0])
    return w, float(rec[w, 0]), int(rec[w, 1])
