"""Multi-GPU plumbing: one process per GPU, population sharded by rows.

Takes the place of the reference's MPI backend (stochopy/optimize/_common.py:45-72: every
rank runs the optimiser, rank 0's candidates are broadcast, fitness is summed with Allreduce).
Here each GPU owns ``popsize / world`` rows for the whole run; per generation the only traffic is
an (n+2)-double record per rank -- [best f, global row, best row] -- after which every rank finalises
the same global best locally.  The population never moves.  Two transports with identical results:
``World.all_gather_records`` (ONE all-gather over RCCL; ``torch.distributed`` backend "nccl" on ROCm =
RCCL over xGMI) and ``PeerExchange`` (DE: the generation kernel writes the record into the peers'
IPC-mapped HBM itself; the process group is only used at set-up).

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


def shard_rows(popsize, world):
    """Rows per shard: ceil(popsize / world).  Every rank but the last owns exactly this many."""
    return -(-int(popsize) // int(world))


def shard_bounds(popsize, world, rank):
    """Rows [row0, row0 + count) of rank `rank`: blocks of ceil(popsize / world) rows in rank order, the LAST rank takes what
    is left -- any popsize is served, as the reference's MPI loop serves any (stochopy/optimize/_common.py:64-65 strides
    `range(rank, len(x), size)`; here blocks, so that a row's owner is `row // shard_rows` and a gather of equal-sized,
    padded blocks holds the population in its first `popsize` rows).  Every rank needs at least one row."""
    c = shard_rows(popsize, world)
    last = popsize - (world - 1) * c
    if last < 1:
        raise ValueError(f"popsize={popsize} over workers={world}: blocks of {c} rows leave the last rank {last} row(s); "
                         "use fewer workers or a larger population")
    row0 = rank * c
    return row0, min(c, popsize - row0)


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

    def shard_rows(self, popsize):
        return shard_rows(popsize, self.size)

    # Graph captures that contain collectives of this group.  torch's ProcessGroupNCCL runs a watchdog thread that polls
    # the completion events of outstanding collectives every 100 ms.  If it polls while this thread captures on the
    # stream those events were recorded on, HIP answers hipErrorCapturedEvent ("operation not permitted on an event last
    # recorded in a capturing stream"), the watchdog throws and the process aborts: 1 in ~20 captures in round 1.
    # Round 3 measured (tools/stress_capture.sh, gpurun_out/r3e): capture_error_mode="thread_local" alone does NOT cure
    # it -- 4 aborts in 40 fresh processes -- because the error is raised for the EVENT, whatever the capture mode.  What
    # cures it is an empty watchdog list while the capture runs.  So: drain the stream, then wait until the watchdog has
    # RETIRED every work (read from torch's flight recorder, no guessing); only if the recorder is unavailable, give the
    # watchdog three of its periods (what round 2 always did: 0 aborts in 135 captures).  thread_local mode is kept: it
    # removes the other, documented hazard (any unsafe call from another thread during a global-mode capture).
    CAPTURE_MODE = "thread_local"

    @staticmethod
    def _watchdog_idle():
        """True / False: every collective in the flight recorder is retired / some are not; None: cannot tell."""
        try:
            import pickle

            from torch._C._distributed_c10d import _dump_nccl_trace

            entries = pickle.loads(_dump_nccl_trace(includeCollectives=True, includeStackTraces=False,
                                                    onlyActive=False)).get("entries", [])
            if not entries or any("retired" not in e for e in entries):
                return None
            return all(bool(e["retired"]) for e in entries)
        except Exception:  # noqa: BLE001  (recorder disabled / other torch version)
            return None

    def quiesce_for_capture(self, ctx):
        """Call right before capturing collectives of this group into a graph (see above)."""
        ctx.sync()
        if self.backend != "nccl":
            return
        import time

        deadline = time.perf_counter() + 2.0
        while True:
            idle = self._watchdog_idle()
            if idle is None:
                time.sleep(0.35)
                return
            if idle:
                return
            if time.perf_counter() > deadline:  # the recorder never reported an empty list: fall back to the waiting rule
                time.sleep(0.35)
                return
            time.sleep(0.01)

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

    def all_gather_rows(self, local, out):
        """out[(total rows, ...)] <- every rank's local[(rows, ...)] in rank order (CMA-ES candidates / fitness).  Shards of
        ceil(total / world) rows, the last one short (shard_bounds): the blocks then travel padded to equal size and the
        population is the first `total` rows of what arrives."""
        import torch

        total, c = int(out.shape[0]), shard_rows(out.shape[0], self.size)
        if c * self.size != total:
            key = (tuple(out.shape[1:]), out.dtype, out.device, c)
            pads = getattr(self, "_pads", None)
            if pads is None:
                pads = self._pads = {}
            if key not in pads:
                pads[key] = (torch.zeros((c,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device),
                             torch.empty((c * self.size,) + tuple(out.shape[1:]), dtype=out.dtype, device=out.device))
            mine, everyone = pads[key]
            mine[:local.shape[0]].copy_(local)
            self.all_gather_rows(mine, everyone)
            out.copy_(everyone[:total])
            return
        if self.backend == "nccl":
            self.dist.all_gather_into_tensor(out.view(-1), local.reshape(-1), group=self.group)
            return
        import torch

        host = local.detach().cpu().reshape(-1)
        gathered = [torch.empty_like(host) for _ in range(self.size)]
        self.dist.all_gather(gathered, host, group=self.group)
        out.view(-1).copy_(torch.cat(gathered).to(out.device))

    def all_gather_object(self, obj):
        """Small host objects (set-up only: IPC handles, flags)."""
        out = [None] * self.size
        self.dist.all_gather_object(out, obj, group=self.group)
        return out

    def barrier(self):
        self.dist.barrier(group=self.group)

    def all_agree(self, flag):
        """True iff `flag` is true on every rank."""
        return all(self.all_gather_object(bool(flag)))

    def max_over_ranks(self, value):
        import torch

        t = torch.tensor([float(value)], dtype=torch.float64)
        if self.backend == "nccl":
            t = t.cuda()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        return float(t.item())


class PeerExchange:
    """Exchange buffers for the peer-exchange generation kernels (include/stochopy_hip.h, sx_xchg_*).

    Every rank allocates one uncached buffer in its HBM, exports it (HIP IPC), maps every peer's buffer and
    runs the transport self-test.  The process group is only used here, at set-up, to pass the 64-byte handles
    around and to agree on the outcome; afterwards the generation kernels write into the peers' memory
    themselves.  ``negotiate`` never leaves ranks in different modes: either all of them get a working
    exchange or all of them get ``None`` (and the reason), and the caller falls back to the RCCL path.
    """

    def __init__(self):
        self.own = None
        self.opened = []
        self.args = None
        self.error = None
        self.pop_own = None
        self.pop_opened = []

    @classmethod
    def negotiate(cls, ctx, world, n, timeout_s=20.0, probe_rounds=8, probe_timeout_s=3.0):
        import ctypes as C

        from . import _lib

        L = ctx.L
        px = cls()
        px.ctx, px.world, px.n = ctx, world, n
        why = None
        handle = None
        if world.size > _lib.SX_MAX_PEERS:
            why = f"more than {_lib.SX_MAX_PEERS} ranks"
        else:
            own = C.c_void_p()
            hbuf = C.create_string_buffer(_lib.SX_IPC_HANDLE_BYTES)
            if L.sx_xchg_alloc(L.sx_xchg_bytes(world.size, n), C.byref(own), hbuf) == 0:
                px.own = own
                handle = hbuf.raw
            else:
                why = "alloc: " + L.sx_last_error().decode()
        handles = world.all_gather_object(handle)
        if any(h is None for h in handles):
            px.close()
            return None, why or "a peer could not allocate / export its exchange buffer"
        # map the peers
        a = _lib.SxXchgArgs()
        a.world, a.rank = world.size, world.rank
        ok = True
        for r, h in enumerate(handles):
            if r == world.rank:
                a.peer[r] = px.own.value
                continue
            p = C.c_void_p()
            if L.sx_xchg_open(h, C.byref(p)) != 0:
                ok, why = False, "open: " + L.sx_last_error().decode()
                break
            px.opened.append(p)
            a.peer[r] = p.value
        t = ctx.zeros((1,), dtype=_torch_int32())
        px.error = t
        a.error = t.data_ptr()
        px.relay = ctx.zeros((int(L.sx_xchg_relay_bytes(n)) // 8,))  # ordinary HBM (zero = no tag yet)
        a.relay = px.relay.data_ptr()
        # the two memsets above ran on torch's CURRENT stream; the probe and the generation kernels use the engine
        # stream, which is not ordered against it: a late memset must not clear a relay tag or the error word
        _device_torch().cuda.current_stream(ctx.device).synchronize()
        if not world.all_agree(ok):
            px.close()
            return None, why or "a peer could not map this rank's exchange buffer"
        # transport self-test (short timeout), then the run's own timeout
        a.timeout_ticks = int(probe_timeout_s * 1e8)
        rc = L.sx_xchg_probe(C.byref(a), n, probe_rounds, ctx.stream_ptr)
        if not world.all_agree(rc == 0):
            px.close()
            return None, ("probe: words did not arrive intact" if rc == 1 else
                          "probe failed on a peer" if rc == 0 else "probe: " + L.sx_last_error().decode())
        a.timeout_ticks = int(timeout_s * 1e8)
        px.args = a
        return px, None

    def share_population(self, rows, n, total=None):
        """Global donors: this rank's two population buffers (rows, n) in memory every peer maps, so that the
        generation kernels can read donor rows from their owners over xGMI.  Returns the two tensors (views of
        one IPC-exported allocation).  Collective; raises on every rank if any rank fails.  `total`: the whole
        population (default world * rows); with a short last shard every rank's second buffer starts behind ITS rows."""
        import ctypes as C

        from . import _lib

        L, world, ctx = self.ctx.L, self.world, self.ctx
        t = __import__("torch")
        nbytes = 2 * rows * n * 8
        own = C.c_void_p()
        hbuf = C.create_string_buffer(_lib.SX_IPC_HANDLE_BYTES)
        ok = L.sx_pop_alloc(nbytes, C.byref(own), hbuf) == 0
        why = None if ok else "alloc: " + L.sx_last_error().decode()
        handles = world.all_gather_object(hbuf.raw if ok else None)
        if any(h is None for h in handles):
            if ok:
                L.sx_xchg_free(own)
            raise RuntimeError(f'donors="global" is not available: {why or "a peer could not export its population"}')
        self.pop_own = own
        bases = []
        for r, h in enumerate(handles):
            if r == world.rank:
                bases.append(own.value)
                continue
            p = C.c_void_p()
            if L.sx_xchg_open(h, C.byref(p)) != 0:
                ok, why = False, "open: " + L.sx_last_error().decode()
                break
            self.pop_opened.append(p)
            bases.append(p.value)
        if not world.all_agree(ok):
            raise RuntimeError(f'donors="global" is not available: {why or "a peer could not map this population"}')
        a = self.args
        total = rows * world.size if total is None else int(total)
        for r, b in enumerate(bases):
            a.pop0[r] = b
            a.pop1[r] = b + shard_bounds(total, world.size, r)[1] * n * 8
        a.shard_rows = shard_rows(total, world.size)  # owner of global row g: g // shard_rows
        a.global_rows = total

        class _Mem:  # zero-copy view of the exported allocation for torch
            __cuda_array_interface__ = {"shape": (2, rows, n), "typestr": "<f8", "data": (own.value, False),
                                        "version": 2, "strides": None}

        self._pop_holder = _Mem()
        both = t.as_tensor(self._pop_holder, device=ctx.device)
        return both[0], both[1]

    def failed(self):
        """True if a wait inside a kernel timed out (synchronises)."""
        self.ctx.sync()
        return bool(int(self.error.cpu()[0]))

    def read_record(self, parity, src):
        """Host copy of slot[parity][src] of this rank's buffer: (f, global row, row)."""
        import ctypes as C

        from . import _lib

        rec = np.empty(self.n + 2)
        _lib.check(self.ctx.L.sx_xchg_read_record(C.byref(self.args), self.n, parity, src,
                                                  rec.ctypes.data_as(C.c_void_p), self.ctx.stream_ptr),
                   "sx_xchg_read_record")
        return float(rec[0]), int(rec[1:2].view(np.int64)[0]), rec[2:].copy()

    def close(self):
        L = self.ctx.L
        for p in self.opened:
            L.sx_xchg_close(p)
        self.opened = []
        if self.own is not None:
            L.sx_xchg_free(self.own)
            self.own = None
        for p in self.pop_opened:
            L.sx_xchg_close(p)
        self.pop_opened = []
        if self.pop_own is not None:  # the tensors handed out by share_population die with it
            L.sx_xchg_free(self.pop_own)
            self.pop_own = None
        self.args = None


def _device_torch():
    import torch

    return torch


def _torch_int32():
    import torch

    return torch.int32


def best_of_records(records):
    """Host restatement of the device rule (sx_gather_finalize): first minimum by (f, global row).
    records: (world, n+2) array-like.  Returns (winner rank, f, global row)."""
    rec = np.asarray(records)
    order = np.lexsort((rec[:, 1], rec[:, 0]))
    w = int(order[0])
    return w, float(rec[w, 0]), int(rec[w, 1])
