"""Differential Evolution front end + generation loop for ``backend="hip"``.

Reference: stochopy/optimize/de/_de.py:13-173 (``minimize``: signature, defaults,
validation, sync rule, seeding) and :176-301 (``de`` loop: initial population,
per-generation draw order, return_all bookkeeping, callback, result).  The
per-generation work -- de_sync (:314-351), the strategies (de/_strategy.py),
``Random`` (de/_constraints.py), selection_sync (_common.py:123-160) and the
objective -- is ONE fused HIP kernel (csrc/sx_de.hip) plus a one-workgroup
best/termination kernel.
"""
import ctypes as C
import os
import warnings

import numpy as np

from .. import _device, _lib, _rng
from . import _common
from ._helpers import OptimizeResult, register

_CAPTURE_MODE = "thread_local"  # see parallel.World.CAPTURE_MODE: torch's NCCL watchdog may poll events while we capture

__all__ = ["minimize"]


def minimize(
    fun,
    bounds,
    x0=None,
    args=(),
    maxiter=100,
    popsize=10,
    mutation=0.5,
    recombination=0.9,
    strategy="best1bin",
    seed=None,
    xtol=1.0e-8,
    ftol=1.0e-8,
    constraints=None,
    updating="immediate",
    workers=1,
    backend=None,
    return_all=False,
    verbosity=1.0,
    callback=None,
    rng=None,
    exchange=None,
    donors=None,
    strict_updating=None,
    host_workers=None,
    host_backend=None,
):
    """Minimize an objective function using Differential Evolution on MI355X.

    Parameters are those of the reference (de/_de.py:13-33); ``backend`` must be
    ``"hip"`` (default), ``workers`` is the number of GPUs, and ``rng`` selects the
    random-draw source: ``"numpy-legacy"`` (default; the reference's stream, so the
    same seed gives the reference's result) or ``"philox"`` (in-kernel counter-based
    draws, the throughput mode).  ``updating="immediate"`` (the reference's default) runs de_async
    (de/_de.py:354-391) as one ordered sweep per generation on one GPU -- same seed, same result as the
    reference's default call -- whenever that is possible (``workers=1``, a factory objective); otherwise the run
    is deferred, as with a parallel backend of the reference (de/_de.py:142-145), and says so in a warning.
    ``updating="deferred"`` is the throughput mode (whole generations in parallel); ``strict_updating=False``
    forces it silently, ``True`` insists on the sweep.  ``exchange`` (``workers > 1`` only) picks how the
    per-generation global best travels between GPUs: ``"p2p"`` (the generation kernel writes its shard's
    record straight into the peers' HBM over xGMI), ``"rccl"`` (one all-gather per generation) or ``None`` /
    ``"auto"`` (p2p if its self-test passes on every rank, else rccl); both give identical results.
    ``donors`` (``workers > 1``): ``"shard"`` (default) draws the donor rows of an individual inside its own
    GPU's shard -- an island model with a shared global best, no population traffic; ``"global"`` draws them over
    the whole population like the reference does (de/_de.py:304-311) and reads them from their owners' HBM over
    xGMI inside the generation kernel: the result of the unsharded run, at the price of remote row reads
    (needs the peer exchange).
    """
    fun_id = _common.resolve_objective(fun, args, workers, backend, host_workers, host_backend)
    lower, upper = _common.as_bounds(bounds)
    if x0 is not None:
        if np.ndim(x0) != 2 or np.shape(x0)[1] != len(bounds):
            raise ValueError()
    if popsize < 2:
        raise ValueError()
    if x0 is not None and len(x0) != popsize:
        raise ValueError()
    if not 0.0 <= mutation <= 2.0:
        raise ValueError()
    if not 0.0 <= recombination <= 1.0:
        raise ValueError()
    if updating not in {"immediate", "deferred"}:
        raise ValueError()
    if strategy not in _lib.DE_STRATEGIES:
        raise KeyError(strategy)
    if constraints not in (None, "Random"):
        raise KeyError(constraints)
    if callback is not None and not hasattr(callback, "__call__"):
        raise ValueError()
    _common.resolve_backend(backend, fun_id)
    rng = _common.resolve_rng(rng)
    workers = _common.resolve_workers(workers, fun_id)
    if popsize - 1 < _lib.DE_DONORS[strategy]:
        raise ValueError()

    run = _DeRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(mutation), float(recombination),
                 strategy, constraints, float(xtol), float(ftol), bool(return_all), float(verbosity), callback, rng,
                 seed, workers, exchange=exchange, donors=donors,
                 immediate=_common.resolve_updating(updating, strict_updating, workers, fun_id, len(lower)))
    return run.result()


class _DeRun:
    GRAPH_CHUNK = 50

    def __init__(self, fun_id, lower, upper, x0, maxiter, P, F, CR, strategy, constraints, xtol, ftol, return_all,
                 verbosity, callback, rng, seed, workers, autorun=True, exchange=None, donors=None, immediate=False):
        self.fun_id, self.lower, self.upper = fun_id, lower, upper
        # a caller-supplied objective (factory.batched) cannot be fused: propose -> fun -> select
        self.external = None if isinstance(fun_id, int) else fun_id
        self.maxiter, self.P, self.n = maxiter, P, len(lower)
        self.F, self.CR, self.strategy, self.constraints = F, CR, strategy, constraints
        self.xtol, self.ftol = xtol, ftol
        self.return_all, self.verbosity, self.callback = return_all, verbosity, callback
        self.rng, self.seed = rng, seed
        self.k = _lib.DE_DONORS[strategy]
        self.world = None
        self.Ptotal = P
        self.row0 = 0
        self.immediate = immediate  # de_async: one sequential sweep per generation (csrc/sx_async.hip)
        if workers != 1 and rng != "philox":
            workers = _common.replicated_workers("de", workers, 'rng="numpy-legacy" replays ONE host stream in the order of the '
                                                 'whole population (rng="philox" shards: draws keyed by the global row)')
        if workers != 1 or os.environ.get("SX_FORCE_SHARDED") == "1":  # the env switch lets a 1-rank group
            from ..parallel import require_world                         # exercise the exchange path (tests)

            self.world = require_world(workers)
            if rng != "philox":
                raise ValueError('a sharded run needs rng="philox" (draws keyed by the global row; see parallel.py)')
            self.row0, self.P = self.world.shard(P)  # this rank's rows; self.P is the LOCAL population from here on
            if immediate:
                raise ValueError("immediate updating is a single-GPU sweep")
        if immediate and self.external is not None:
            raise ValueError("immediate updating evaluates individuals one by one inside the sweep kernel: "
                             "only the factory objectives can do that")
        if donors not in (None, "shard", "global"):
            raise ValueError('donors must be "shard" or "global"')
        self.global_donors = donors == "global" and self.world is not None
        if self.world is not None and self.world.size > 1 and donors is None and not _DeRun._warned_island:
            # The reference's invariant is "the backend does not change the result for a seed" (its tests/helpers.py:28-36,
            # stochopy/optimize/_common.py:58-72).  The DEFAULT here breaks it knowingly: SURVEY.md section 8e / the north star
            # shard the population "embarrassingly", i.e. donors come from the rank's own rows.  Say so once; donors="shard"
            # (chosen, no warning) and donors="global" (the unsharded run, bit for bit) are the explicit spellings.
            _DeRun._warned_island = True
            warnings.warn('stochopy_amd: de with workers > 1 draws donors from each rank\'s own rows by default (an island model '
                          'with a shared global best): the result for a seed then depends on the number of workers, unlike the '
                          'reference\'s parallel backends.  options["donors"]="global" reproduces the workers=1 run bit for bit '
                          '(donor rows are read over xGMI; needs the peer exchange); options["donors"]="shard" keeps this mode '
                          'without the warning.  (Shown once.)', UserWarning, stacklevel=4)
        if self.world is not None:
            # (every rank checks the SMALLEST shard -- the last rank's, parallel.shard_bounds -- so that all of them raise, or none)
            smallest = self.Ptotal - (self.world.size - 1) * self.world.shard_rows(self.Ptotal)
            if smallest < 2:
                raise ValueError(f"popsize {self.Ptotal} over {self.world.size} ranks leaves {smallest} row(s) on the last GPU: "
                                 "a shard needs at least 2")
            if (self.Ptotal if self.global_donors else smallest) - 1 < self.k:
                raise ValueError(f"strategy {strategy} draws {self.k} donors: too few rows "
                                 f"({'population' if self.global_donors else 'smallest shard'} of "
                                 f"{self.Ptotal if self.global_donors else smallest})")
        self.x0 = x0
        # single GPU + in-kernel draws + nothing to report per generation: one kernel per generation
        # ("chained finalize", include/stochopy_hip.h sx_de_chain_launch)
        # -- every wavefront re-reduces the per-workgroup records, so only while those are few (<= 512)
        # Rows of more than wide_from() (2048) elements take the one-workgroup-per-row kernels (csrc/sx_wide.hip, two kernels per
        # generation); the peer exchange and the global-donor gathers live in the chained kernel, which serves rows of up to
        # NARROW_DIM (4096) elements: a run that ASKS for them on rows of 2049 ... 4096 elements gets the wavefront-per-row
        # kernels for its duration (sx_set_wide_from; restored by close()).  exchange="auto" keeps the faster wide kernels and
        # the all-gather there.
        self._wide_from_prev = None
        needs_narrow = (self.world is not None and self.external is None
                        and (donors == "global" or (exchange or os.environ.get("SX_EXCHANGE")) == "p2p"))
        if needs_narrow and _lib.wide_from() < self.n <= _lib.NARROW_DIM:
            self._wide_from_prev = int(_lib.lib().sx_set_wide_from(_lib.NARROW_DIM))
        try:
            npart = int(_lib.lib().sx_num_partials(self.P, self.n))
            self.wide = self.n > _lib.wide_from()
            self.chain = (rng == "philox" and self.world is None and callback is None and not return_all
                          and npart <= 512 and not immediate and self.external is None and not self.wide)
            self.launches = 0
            self.ctx = _device.Context()
            # multi-GPU: the chained kernel with the peer exchange in its prologue, if the transport checks out
            self.px = None
            self.exchange = None
            if self.world is not None:
                exchange = exchange or os.environ.get("SX_EXCHANGE") or "auto"
                if exchange not in ("auto", "p2p", "rccl"):
                    raise ValueError('exchange must be "auto", "p2p" or "rccl"')
                self.exchange, self.exchange_note = "rccl", None
                if self.external is not None:
                    if exchange == "p2p" or self.global_donors:
                        raise ValueError("a caller-supplied objective runs between kernels: the peer exchange lives "
                                         'inside the fused generation kernel (use exchange="rccl", donors="shard")')
                    exchange = "rccl"
                if self.wide:  # the peer exchange lives in the chained kernel, which serves rows of <= NARROW_DIM elements
                    if exchange == "p2p" or self.global_donors:
                        raise ValueError(f"rows of {self.n} elements (> {_lib.NARROW_DIM}) exchange the global best with one "
                                         'all-gather per generation (exchange="rccl", donors="shard")')
                    exchange = "rccl"
                if exchange != "rccl":
                    from ..parallel import PeerExchange

                    timeout = float(os.environ.get("SX_XCHG_TIMEOUT_S", "20"))
                    self.px, self.exchange_note = PeerExchange.negotiate(self.ctx, self.world, self.n, timeout)
                    if self.px is not None:
                        self.exchange, self.chain = "p2p", True
                    elif exchange == "p2p":
                        raise RuntimeError(f'exchange="p2p" is not available: {self.exchange_note}')
                if self.global_donors and self.px is None:
                    raise RuntimeError('donors="global" needs the peer exchange (exchange="p2p"/"auto"): '
                                       f'{self.exchange_note or "it was switched off"}')
            self._graph = None
            self._chain_graphs = {}
            self._tail_seen = {}
            self._ext_graphs, self._ext_graph_note = {}, None
            self._shard_calls = None
            self._rccl_graph = None
            self._rccl_graphs, self._rccl_tail_seen = {}, {}
            self._rccl_graph_note = None
            if autorun:
                t = _device.torch()
                with t.cuda.stream(self.ctx.stream):
                    ok = False
                    try:
                        self._run()
                        ok = True
                    finally:
                        try:
                            if self.px is not None:
                                # Peers may still be reading this rank's exchange / population memory (their last kernels,
                                # remote donor rows): nobody unmaps or frees anything before EVERY rank has drained its
                                # stream.  The meeting point is reached by failing ranks too (it carries a success flag):
                                # a rank whose objective / callback raised makes its peers raise, not hang in a barrier.
                                if ok:
                                    self.ctx.sync()
                                if not self.world.all_agree(ok) and ok:
                                    raise RuntimeError("a peer rank failed during the run (its own exception says why)")
                        finally:
                            self.close()
        except BaseException:
            self._restore_wide_from()
            raise

    def _restore_wide_from(self):
        if self._wide_from_prev is not None:
            _lib.lib().sx_set_wide_from(self._wide_from_prev)
            self._wide_from_prev = None

    def close(self):
        if self._wide_from_prev is not None:
            self.ctx.sync()
            self._restore_wide_from()
        if self._rccl_graph is not None or self._ext_graphs:
            self.ctx.sync()
            self._rccl_graph = None
            self._rccl_graphs = {}
            self._ext_graphs = {}
        if self._graph is not None:
            self.ctx.L.sx_graph_destroy(self._graph)
            self._graph = None
        for g in self._chain_graphs.values():
            self.ctx.L.sx_graph_destroy(g)
        self._chain_graphs = {}
        if self.px is not None:
            self.ctx.sync()
            if self.global_donors:
                self.bufs = None  # views of the shared allocation that px.close() releases
            self.px.close()
            self.px = None

    def read_state(self):
        """Host view of the run: (chained mode) finalise the last generation into state[2], then read it."""
        ctx = self.ctx
        if not self.chain:
            return ctx.read_state(self.state)
        self._chain_launch(self.launches & 1, 1)
        st = ctx.read_state(self.state[16:24])
        if self.px is not None and self.px.failed():
            raise RuntimeError("peer exchange timed out: a rank did not reach the generation the others "
                               "were waiting for (SX_XCHG_TIMEOUT_S)")
        return st

    def _chain_launch(self, parity, finalize_only):
        ctx = self.ctx
        if self.px is not None:
            _lib.check(ctx.L.sx_de_p2p_launch(C.byref(self.args), C.byref(self.px.args), parity, finalize_only,
                                              ctx.stream_ptr), "sx_de_p2p_launch")
        else:
            _lib.check(ctx.L.sx_de_chain_launch(C.byref(self.args), parity, finalize_only, ctx.stream_ptr),
                       "sx_de_chain_launch")

    _warned_island = False
    TAIL_CHUNK = 10  # a second, short graph: runs of fewer than GRAPH_CHUNK generations are replayed too (even: parity)

    def _chain_graph(self, par, size=None):
        size = size or self.GRAPH_CHUNK
        key = (par, size)
        if key not in self._chain_graphs:
            g = C.c_void_p()
            if self.px is not None:
                _lib.check(self.ctx.L.sx_de_p2p_graph_create(C.byref(self.args), C.byref(self.px.args),
                                                             size, par, C.byref(g)), "sx_de_p2p_graph_create")
            else:
                _lib.check(self.ctx.L.sx_de_chain_graph_create(C.byref(self.args), size, par, C.byref(g)),
                           "sx_de_chain_graph_create")
            self._chain_graphs[key] = g
        return self._chain_graphs[key]

    def prepare_graphs(self):
        """Instantiate the hipGraph(s) up front (otherwise the first full chunk pays for it)."""
        if self.world is not None and not self.chain:
            return
        if self.chain:
            self._chain_graph(0)  # chunk sizes are even: replays always start at parity 0 unless eager launches intervene
            self._chain_graph(0, self.TAIL_CHUNK)
        elif self.rng == "philox" and self._graph is None:
            g = C.c_void_p()
            _lib.check(self.ctx.L.sx_de_graph_create(C.byref(self.args), self.GRAPH_CHUNK, C.byref(g)),
                       "sx_de_graph_create")
            self._graph = g

    @staticmethod
    def plan_chain(ngen, launches, seen, chunk, tail):
        """How `ngen` generations of the chained kernel are enqueued after `launches` earlier ones: a list of graph
        lengths (0 = one eager launch).  Whole `chunk`-generation graphs first; the rest as ONE graph of exactly that
        (even) length once the same (parity, length) has been asked for before -- every replay costs about as much as
        one generation of the metric shape, so stepping in blocks of 20 becomes one replay per block instead of two of
        the `tail`-generation graph, while a length seen once (the tail of a single minimize() call) is not worth an
        instantiation; at most chunk/2 lengths x 2 parities are ever instantiated.  `seen` counts the requests."""
        plan = [chunk] * (ngen // chunk)
        ngen -= chunk * len(plan)
        size = ngen & ~1
        if size > tail:
            key = ((launches + sum(plan)) & 1, size)
            seen[key] = seen.get(key, 0) + 1
            if seen[key] >= 2:
                plan.append(size)
                ngen -= size
        plan += [tail] * (ngen // tail)
        ngen %= tail
        return plan + [0] * ngen

    def _enqueue_chain(self, ngen):
        ctx = self.ctx
        for size in self.plan_chain(ngen, self.launches, self._tail_seen, self.GRAPH_CHUNK, self.TAIL_CHUNK):
            par = self.launches & 1
            if size == 0:
                self._chain_launch(par, 0)
                self.launches += 1
            else:
                _lib.check(ctx.L.sx_graph_launch(self._chain_graph(par, size), ctx.stream_ptr), "sx_graph_launch")
                self.launches += size

    def _sharded_generation(self):
        """One generation on this rank's shard + the global-best exchange (parallel.py): two host calls into
        the library around one all-gather."""
        ctx, L = self.ctx, self.ctx.L
        if self._shard_calls is None:  # argument objects built once: the loop below is host-bound
            p = _device.ptr
            self._shard_calls = (C.byref(self.args), p(self.record), p(self.records), p(self.gbest), p(self.state),
                                 ctx.stream_ptr)
        a, rec, recs, gb, st, sp = self._shard_calls
        if L.sx_de_shard_generation(a, rec, sp) != 0:
            _lib.check(-1, "sx_de_shard_generation")
        self.world.all_gather_records(self.record, self.records)
        if L.sx_gather_finalize(recs, self.world.size, self.n, gb, st, self.maxiter, self.xtol, self.ftol, sp) != 0:
            _lib.check(-1, "sx_gather_finalize")

    def _capture_sharded_chunk(self, size=None):
        """Capture `size` (default GRAPH_CHUNK) generations of the "rccl" transport (kernels + all-gathers) into one graph.
        False when that is not possible (gloo staging goes through the host; SX_RCCL_GRAPH=0; a failed capture):
        the caller then launches generation by generation -- the same sequence of collectives either way, so
        ranks need not agree on which form they use."""
        size = size or self.GRAPH_CHUNK
        if size in self._rccl_graphs:
            self._rccl_graph = self._rccl_graphs[size]
            return True
        if (self._rccl_graph_note is not None or self.world.backend != "nccl"
                or os.environ.get("SX_RCCL_GRAPH") == "0"):
            return False
        t = _device.torch()
        try:
            self.world.quiesce_for_capture(self.ctx)
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g, stream=self.ctx.stream, capture_error_mode=_CAPTURE_MODE):
                for _ in range(size):
                    self._sharded_generation()
            self._rccl_graphs[size] = self._rccl_graph = g
            return True
        except Exception as e:  # capture is an optimisation, never a requirement
            self._rccl_graph_note = f"graph capture of the rccl path failed: {e}"
            return False

    def enqueue(self, ngen):
        """Enqueue `ngen` generations on the engine stream without any host synchronisation.

        Philox mode only.  Full chunks replay one instantiated hipGraph (two kernel nodes per
        generation); the remainder is launched eagerly.  Generations after convergence are no-ops.
        """
        ctx = self.ctx
        if self.chain:
            self._enqueue_chain(ngen)
            return
        if self.world is not None:
            # "rccl" transport: kernels + the all-gather of a chunk of generations captured once into a graph
            # (RCCL collectives are capturable) and replayed, so the host is off the per-generation path
            while ngen >= self.GRAPH_CHUNK and self._capture_sharded_chunk():
                self._rccl_graph.replay()
                ngen -= self.GRAPH_CHUNK
            # a remainder that keeps coming back (stepping in blocks of K < GRAPH_CHUNK generations: the driver's bench uses
            # K = 20) gets a graph of its own the second time it is asked for, as the chained kernel's tails do (plan_chain);
            # every rank sees the same requests, so every rank captures at the same call
            if ngen >= 4:
                self._rccl_tail_seen[ngen] = self._rccl_tail_seen.get(ngen, 0) + 1
                if (self._rccl_tail_seen[ngen] >= 2 and len(self._rccl_graphs) < 4) or ngen in self._rccl_graphs:
                    if self._capture_sharded_chunk(ngen):
                        self._rccl_graph.replay()
                        ngen = 0
            for _ in range(ngen):
                self._sharded_generation()
            return
        while ngen >= self.GRAPH_CHUNK:
            if self._graph is None:
                g = C.c_void_p()
                _lib.check(ctx.L.sx_de_graph_create(C.byref(self.args), self.GRAPH_CHUNK, C.byref(g)),
                           "sx_de_graph_create")
                self._graph = g
            _lib.check(ctx.L.sx_graph_launch(self._graph, ctx.stream_ptr), "sx_graph_launch")
            ngen -= self.GRAPH_CHUNK
        for _ in range(ngen):
            _lib.check(ctx.L.sx_de_generation(C.byref(self.args), 1, ctx.stream_ptr), "sx_de_generation")

    # ------------------------------------------------------------------ setup
    def _setup(self):
        ctx, P, n = self.ctx, self.P, self.n
        t = _device.torch()
        self.stream = _rng.make_init_stream(self.rng, self.seed)
        d_bounds = ctx.upload_async(np.concatenate([self.lower, self.upper]))  # (a blocking upload costs ~0.2 ms)
        self.d_lower, self.d_upper = d_bounds[:n], d_bounds[n:]
        # generation g lives in bufs[g & 1]; the initial population is generation 1
        if self.global_donors:  # buffers every peer maps: donor rows are read from their owners over xGMI
            self.bufs = list(self.px.share_population(P, n, self.Ptotal))
        else:
            self.bufs = [ctx.empty((P, n)), ctx.empty((P, n))]
        if self.x0 is None and self.rng == "philox":
            # in-kernel draws: the Latin hypercube is drawn on the device too, every rank its own rows (_rng.py)
            _rng.philox_latin_hypercube(ctx, self.bufs[1], self.row0, self.Ptotal, self.d_lower, self.d_upper, self.seed)
        else:
            if self.x0 is not None:
                X0 = np.array(self.x0, dtype=np.float64)
            else:  # the reference's stream: sequential by construction, built whole (numpy-legacy runs on one GPU)
                X0 = self.stream.latin_hypercube(self.Ptotal, n, self.lower, self.upper)
            if self.world is not None:
                X0 = np.ascontiguousarray(X0[self.row0 : self.row0 + P])
            self.bufs[1].copy_(ctx.upload(X0))
        self.fit = ctx.empty((P,))
        self.candfit = ctx.empty((P,))
        npart = int(ctx.L.sx_num_partials(P, n))
        self.part_f = ctx.empty((npart,))
        self.part_i = ctx.empty((npart,), dtype=t.int64)
        # initial evaluation and best (de/_de.py:212-218)
        _common.evaluate_rows(ctx, self.fun_id, self.bufs[1], n, self.fit)
        self.candfit.copy_(self.fit)
        # One GPU, chained kernels, nobody watching generation 1 (round 3): the initial best never visits the host -- the
        # device argmin writes it straight into record 0 of the chain's first record set, from which launch 0 "finalises"
        # generation 1.  (Before: two blocking reads in the middle of the set-up and three more uploads, ~0.35 ms of the
        # 0.8 ms a whole minimize() call costs before its first generation.)
        quiet_start = (self.chain and self.world is None and self.px is None and self.callback is None
                       and not self.return_all and self.external is None)
        if quiet_start:
            rec_f = t.full((2, npart), float("inf"), dtype=t.float64, device=ctx.device)
            rec_i = t.full((2, npart), np.iinfo(np.int64).max, dtype=t.int64, device=ctx.device)
            _lib.check(ctx.L.sx_argmin(_device.ptr(self.fit), P, _device.ptr(self.part_f), _device.ptr(self.part_i),
                                       npart, _device.ptr(rec_i), _device.ptr(rec_f), ctx.stream_ptr), "sx_argmin")
            g, gfit0 = 0, 0.0  # placeholders: the host learns the best with its first look at the state
            self.gbest = None
        else:
            out_i = ctx.empty((1,), dtype=t.int64)
            out_f = ctx.empty((1,))
            _lib.check(ctx.L.sx_argmin(_device.ptr(self.fit), P, _device.ptr(self.part_f), _device.ptr(self.part_i),
                                       npart, _device.ptr(out_i), _device.ptr(out_f), ctx.stream_ptr), "sx_argmin")
            g = int(out_i.cpu()[0])
            gfit0 = float(out_f.cpu()[0])
            self.gbest = self.bufs[1][g].clone()
        if self.world is not None and self.px is None:  # initial global best: one record exchange, settled on the host
            from ..parallel import best_of_records

            self.record = ctx.empty((n + 2,))
            self.records = ctx.empty((self.world.size, n + 2))
            self.record[0] = gfit0
            self.record[1] = float(self.row0 + g)
            self.record[2:].copy_(self.gbest)
            self.world.all_gather_records(self.record, self.records)
            wbest, gfit0, g = best_of_records(self.records.cpu().numpy())
            self.gbest.copy_(self.records[wbest, 2:])
        st = _lib.SxState(it=1, gbidx=g, gfit=gfit0, dx=0.0, status=_lib.SX_STATUS_NONE, done=0)
        if self.chain:
            # state[3] + records[2][npart]: launch 0 (parity 0) first "finalises" generation 1 from records[0]
            s0 = _lib.SxState(it=0, gbidx=g, gfit=gfit0, dx=0.0, status=_lib.SX_STATUS_NONE, done=0)
            raw = np.frombuffer(bytes(s0) + bytes(st) + bytes(st), dtype=np.int64).copy()
            self.state = ctx.upload_async(raw)
            if quiet_start:
                self.part_f, self.part_i = rec_f, rec_i
            else:
                pf = np.full((2, npart), np.inf)
                pi = np.full((2, npart), np.iinfo(np.int64).max, dtype=np.int64)
                pf[0, 0], pi[0, 0] = gfit0, g
                self.part_f = ctx.upload(pf)
                self.part_i = ctx.upload(pi)
        else:
            self.state = ctx.upload(np.frombuffer(bytes(st), dtype=np.int64).copy())
        key0, key1 = _rng.philox_key(self.seed) if self.rng == "philox" else (0, 0)
        a = _lib.SxDeArgs()
        a.buf0, a.buf1 = self.bufs[0].data_ptr(), self.bufs[1].data_ptr()
        if self.immediate:  # ONE population, updated in place (the initial one sits in bufs[1])
            a.buf0 = a.buf1
        a.fit, a.candfit = self.fit.data_ptr(), self.candfit.data_ptr()
        a.lower, a.upper, a.state = self.d_lower.data_ptr(), self.d_upper.data_ptr(), self.state.data_ptr()
        a.part_f, a.part_i = self.part_f.data_ptr(), self.part_i.data_ptr()
        # chained mode reads the best row straight from the population (row state.gbidx): no copy to maintain
        a.gbest = None if self.chain else self.gbest.data_ptr()
        a.P, a.ld, a.row0, a.n = P, n, self.row0, n
        a.fun_id, a.strategy = (self.fun_id if self.external is None else 0), _lib.DE_STRATEGIES[self.strategy]
        if self.external is not None:
            self.cand = ctx.empty((P, n))
            self.cand_f = ctx.empty((P,))
        self.it_enq = 1  # generation the device holds once everything enqueued so far has run
        a.constraints = 1 if self.constraints == "Random" else 0
        a.rng = _lib.SX_RNG_PHILOX if self.rng == "philox" else _lib.SX_RNG_HOST
        a.maxiter = self.maxiter
        a.F, a.CR, a.xtol, a.ftol = self.F, self.CR, self.xtol, self.ftol
        a.key0, a.key1 = key0, key1
        self.args = a
        if self.rng == "numpy-legacy":
            # pinned staging + device buffers for one generation of host draws
            self.h_r1 = t.empty((P, n), dtype=t.float64).pin_memory()
            self.h_don = t.empty((self.k, P), dtype=t.int32).pin_memory()
            self.h_irand = t.empty((P,), dtype=t.int32).pin_memory()
            self.d_r1 = ctx.empty((P, n))
            self.d_don = ctx.empty((self.k, P), dtype=t.int32)
            self.d_irand = ctx.empty((P,), dtype=t.int32)
            a.r1, a.donors, a.irand = self.d_r1.data_ptr(), self.d_don.data_ptr(), self.d_irand.data_ptr()
            if a.constraints:
                self.h_rs = t.empty((P, n), dtype=t.float64).pin_memory()
                self.d_rs = ctx.empty((P, n))
                a.resample = self.d_rs.data_ptr()
        # return_all history (de/_de.py:221-234) kept in HBM, copied out once at the end
        self.st = st
        if self.world is not None and (self.return_all or self.callback is not None):
            self.Xfull = ctx.empty((self.Ptotal, n))
            self.candfull = ctx.empty((self.Ptotal,))
            if self.px is not None:  # generation 1 finalised over all ranks (a finalise-only launch on every rank)
                st = self.read_state()
                self.gbest.copy_(ctx.upload(self._best_row(st)))
        if self.return_all:
            self.nout = int(np.ceil(self.verbosity * self.Ptotal))
            rows = max(self.nout, 1)
            self.xall = ctx.empty((self.maxiter, rows, n))
            self.funall = ctx.empty((self.maxiter, rows))
            if self.nout > 0:
                X1, f1 = self._whole_population(1)
                self.xall[0].copy_(X1[: self.nout])
                self.funall[0].copy_(f1[: self.nout])
            else:
                self.xall[0, 0].copy_(self.gbest if self.world is not None else self.bufs[1][g])
                self.funall[0, 0] = st.gfit
        self.st = st

    # --------------------------------------------------------------- helpers
    def _population(self, it):
        """Device view of generation `it`'s population (P, n)."""
        return self.bufs[1] if self.immediate else self.bufs[it & 1]

    def _whole_population(self, it):
        """(population, candidate fitness) of generation `it` as the caller sees them: with workers > 1 every
        rank gathers all shards (callbacks / return_all only -- the reference's parallel backends also hand the
        whole population to the callback on every rank)."""
        X = self._population(it)
        if self.world is None:
            return X, self.candfit
        self.world.all_gather_rows(X, self.Xfull)
        self.world.all_gather_rows(self.candfit, self.candfull)
        return self.Xfull, self.candfull

    def _record(self, it):
        """return_all bookkeeping for generation `it` (de/_de.py:270-278)."""
        if not self.return_all:
            return
        X, cand = self._whole_population(it)
        if self.nout > 0:
            self.xall[it - 1].copy_(X[: self.nout])
            self.funall[it - 1].copy_(cand[: self.nout])  # candidate fitness, de/_de.py:270-273
        else:
            k = int(cand.argmin())
            self.xall[it - 1, 0].copy_(X[k])
            self.funall[it - 1, 0] = cand[k]

    def _best_row(self, st):
        """The best individual of generation st.it (host copy)."""
        if self.px is not None:  # generation `it` was finalised from the records in slot parity (it-1)&1
            return self.px.read_record((st.it - 1) & 1, int(st.reserved[1]))[2]
        if self.chain:
            return self._population(st.it)[st.gbidx].cpu().numpy()
        return self.gbest.cpu().numpy()

    def _settle_status(self, st):
        """Chained mode stops on `fun <= ftol` with status 1; _common.py:135-140 calls it 0 when the best moved
        by <= xtol.  Both generations are still resident (nothing is produced after `done`)."""
        status = int(st.status)
        if self.chain and status == 1:
            if self.px is not None:  # the previous generation's records sit in the other parity
                prev = self.px.read_record(st.it & 1, int(st.reserved[0]))[2]
            else:
                prev = self._population(st.it - 1)[st.reserved[0]].cpu().numpy()
            if np.linalg.norm(prev - self._best_row(st)) <= self.xtol:
                status = 0
        return status

    def _partial_result(self, st):
        res = OptimizeResult(x=self._best_row(st), fun=st.gfit, nfev=st.it * self.Ptotal, nit=st.it)
        if self.return_all:
            res.update({"xall": self.xall[: st.it].cpu().numpy(), "funall": self.funall[: st.it].cpu().numpy()})
        return res

    def _host_draws(self):
        """One generation of the numpy-legacy stream, in the reference's order."""
        s = self.stream
        s.random(None, out=self.h_r1.numpy())                      # de/_de.py:250
        if self.immediate:  # :376-382, individual by individual: donors, forced index, Random's block
            rs = self.h_rs.numpy() if self.args.constraints else None
            s.de_async_draws(self.P, self.k, self.n, self.h_don.numpy(), self.h_irand.numpy(), self.lower, self.upper, rs)
            self.d_r1.copy_(self.h_r1, non_blocking=True)
            self.d_don.copy_(self.h_don, non_blocking=True)
            self.d_irand.copy_(self.h_irand, non_blocking=True)
            if rs is not None:
                self.d_rs.copy_(self.h_rs, non_blocking=True)
            return
        s.de_donors(self.P, self.k, out=self.h_don.numpy())        # :304-311
        self.h_irand.numpy()[:] = s.randint(self.n, self.P)        # :340
        self.d_r1.copy_(self.h_r1, non_blocking=True)
        self.d_don.copy_(self.h_don, non_blocking=True)
        self.d_irand.copy_(self.h_irand, non_blocking=True)
        if self.args.constraints:
            s.uniform_rows(self.lower, self.upper, self.P, out=self.h_rs.numpy())  # de/_constraints.py:24
            self.d_rs.copy_(self.h_rs, non_blocking=True)

    def _external_generation(self):
        """One generation around the caller's objective: candidates -> fun -> selection -> best/termination
        (csrc/sx_unfused.hip).  Generation g lives in bufs[g & 1]; after convergence every kernel is a no-op."""
        ctx, L, p, n = self.ctx, self.ctx.L, _device.ptr, self.n
        it = self.it_enq
        _lib.check(L.sx_de_propose(C.byref(self.args), p(self.cand), ctx.stream_ptr), "sx_de_propose")
        self.cand_f.copy_(self.external(ctx, self.cand))
        cur, nxt = self.bufs[it & 1], self.bufs[(it + 1) & 1]
        _lib.check(L.sx_rows_select(p(self.cand), n, p(self.cand_f), p(cur), p(nxt), n, p(self.fit), p(self.candfit),
                                    self.P, n, p(self.state), p(self.part_f), p(self.part_i), ctx.stream_ptr),
                   "sx_rows_select")
        npart = int(L.sx_num_partials(self.P, n))
        if self.world is None:
            _lib.check(L.sx_select_finalize(p(self.part_f), p(self.part_i), npart, p(self.bufs[0]), p(self.bufs[1]), n, n,
                                            p(self.gbest), p(self.state), self.maxiter, self.xtol, self.ftol,
                                            ctx.stream_ptr), "sx_select_finalize")
        else:
            _lib.check(L.sx_shard_best(p(self.part_f), p(self.part_i), npart, p(self.bufs[0]), p(self.bufs[1]), n, n,
                                       p(self.state), self.row0, p(self.record), ctx.stream_ptr), "sx_shard_best")
            self.world.all_gather_records(self.record, self.records)
            _lib.check(L.sx_gather_finalize(p(self.records), self.world.size, n, p(self.gbest), p(self.state),
                                            self.maxiter, self.xtol, self.ftol, ctx.stream_ptr), "sx_gather_finalize")
        self.it_enq = it + 1

    EXT_CHUNK = 16  # generations per captured graph around a caller-supplied objective (even: buffer parity)

    def _external_graph(self, parity):
        """EXT_CHUNK generations -- our kernels AND the caller's device objective -- captured once into a graph
        (in-kernel draws).  False when that is not possible (host draws, gloo, SX_EXT_GRAPH=0, or an objective
        that cannot be captured, e.g. one that synchronises): the caller then launches eagerly."""
        if parity in self._ext_graphs:
            return True
        if (self._ext_graph_note is not None or self.rng != "philox" or not getattr(self.external, "capturable", True)
                or (self.world is not None and self.world.backend != "nccl") or os.environ.get("SX_EXT_GRAPH") == "0"):
            return False
        t = _device.torch()
        it0 = self.it_enq
        try:
            self.ctx.sync()
            if self.world is not None:
                self.world.quiesce_for_capture(self.ctx)
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g, stream=self.ctx.stream, capture_error_mode=_CAPTURE_MODE):
                for _ in range(self.EXT_CHUNK):
                    self._external_generation()
            self._ext_graphs[parity] = g
        except Exception as e:  # capture is an optimisation, never a requirement
            self._ext_graph_note = f"graph capture around the objective failed: {e}"
        self.it_enq = it0  # capturing ran nothing
        return parity in self._ext_graphs

    def _enqueue_external(self, ngen):
        while ngen >= self.EXT_CHUNK and self._external_graph(self.it_enq & 1):
            self._ext_graphs[self.it_enq & 1].replay()
            self.it_enq += self.EXT_CHUNK
            ngen -= self.EXT_CHUNK
        for _ in range(ngen):
            self._external_generation()

    def _generation(self):
        """One generation on the engine stream: the fused kernel + best/termination, or the sequential sweep;
        with workers > 1 the shard's generation + the exchange of the global best."""
        ctx = self.ctx
        if self.external is not None:
            self._external_generation()
        elif self.px is not None:
            self._chain_launch(self.launches & 1, 0)
            self.launches += 1
        elif self.world is not None:
            self._sharded_generation()
        elif self.immediate:
            _lib.check(ctx.L.sx_de_async_generation(C.byref(self.args), ctx.stream_ptr), "sx_de_async_generation")
        else:
            _lib.check(ctx.L.sx_de_generation(C.byref(self.args), 1, ctx.stream_ptr), "sx_de_generation")

    # ------------------------------------------------------------------ loop
    def _run(self):
        ctx = self.ctx
        self._setup()
        st = self.st
        if self.callback is not None:
            self.callback(self._whole_population(1)[0].cpu().numpy(), self._partial_result(st))
        # return_all with in-kernel draws: the per-generation history copies (de/_de.py:270-278) are device-side
        # and ordered on the engine stream, so the host need not look at every generation
        record_async = (self.return_all and self.rng == "philox" and self.callback is None and self.nout > 0
                        and self.maxiter > 1)
        stepwise = (self.rng == "numpy-legacy" or self.callback is not None or self.return_all) and not record_async
        while not st.done:
            # maxiter <= 1: the reference still runs one generation before it tests `it >= maxiter`
            remaining = max(self.maxiter - st.it, 1)
            if record_async:
                for j in range(min(remaining, 32)):
                    self._generation()
                    self._record(st.it + 1 + j)  # generations after convergence are no-ops; their slots are cut off
                st = self.read_state()
                self.it_enq = int(st.it)
            elif stepwise:
                if self.rng == "numpy-legacy":
                    self._host_draws()
                self._generation()
                self._record(st.it + 1)
                st = self.read_state()
                self.it_enq = int(st.it)
                if self.callback is not None:
                    self.callback(self._whole_population(st.it)[0].cpu().numpy(), self._partial_result(st))
            elif self.immediate:  # sweeps are long (P sequential individuals): look after every few of them
                for _ in range(min(remaining, 8)):
                    self._generation()
                st = ctx.read_state(self.state)
            elif self.external is not None:  # kernels and the caller's objective, queued on the engine stream
                self._enqueue_external(min(remaining, 4 * self.EXT_CHUNK))
                st = self.read_state()
                self.it_enq = int(st.it)
            elif self.chain and self.px is None:
                st = self._run_chain_ahead(st)
            else:
                # termination is tested on the device every generation; the host looks every <=4 chunks
                self.enqueue(min(remaining, 4 * self.GRAPH_CHUNK))
                st = self.read_state()
        self.st = st
        status = self._settle_status(st)
        res = OptimizeResult(
            x=self._best_row(st),
            success=status >= 0,
            status=status,
            message=_common.messages[status],
            fun=float(st.gfit),
            nfev=int(st.it) * self.Ptotal,
            nit=int(st.it),
        )
        if self.return_all:
            res.update({"xall": self.xall[: st.it].cpu().numpy(), "funall": self.funall[: st.it].cpu().numpy()})
        # the reference works in place on x0 (de/_de.py:208 + _common.py:128-129): mirror that
        if self.world is None and isinstance(self.x0, np.ndarray) and self.x0.dtype == np.float64:
            self.x0[...] = self._population(st.it).cpu().numpy()
        if self.rng == "numpy-legacy":
            self.stream.sync_back()
        ctx.sync()
        # (peer exchange: the one meeting point of all ranks -- success flag included -- is `all_agree` in the caller's
        #  `finally`; a barrier here would pair with a failing rank's all_gather there: mismatched collectives)
        self._res = res

    def _run_chain_ahead(self, st):
        """One GPU, chained kernels: the host looks at the state every four replays as before, but one more replay is
        already queued when it does -- the device never waits for the look (a blocking read of the 64-byte state, ~60 us
        with the re-enqueueing behind it: 0.3 us per generation of a long run).  Termination is decided on the device in every
        generation; a run that stops on ftol leaves at most one replay of no-op launches (<= GRAPH_CHUNK x ~2 us) behind."""
        ctx = self.ctx
        t = _device.torch()
        B, L = 4 * self.GRAPH_CHUNK, self.GRAPH_CHUNK
        pin = t.empty(8, dtype=t.int64).pin_memory()
        ev = t.cuda.Event()
        enq, first = int(st.it), True
        while True:
            n = min(max(self.maxiter - enq, 1 if first else 0), B if first else B - L)
            if n > 0:
                self.enqueue(n)
                enq += n
            self._chain_launch(self.launches & 1, 1)  # finalise the last generation enqueued into state[2] (the host's view)
            with t.cuda.stream(ctx.stream):
                pin.copy_(self.state[16:24], non_blocking=True)
            ev.record(ctx.stream)
            n = min(self.maxiter - enq, L)
            if n > 0:  # the look-ahead: queued before the host waits for the copy
                self.enqueue(n)
                enq += n
            ev.synchronize()
            st = _lib.SxState.from_buffer_copy(pin.numpy().tobytes())
            if st.done:
                return st
            first = False

    def result(self):
        return self._res


register("de", minimize)
