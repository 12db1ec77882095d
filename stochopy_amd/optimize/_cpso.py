"""PSO / CPSO front end + generation loop for ``backend="hip"``.

Reference: stochopy/optimize/cpso/_cpso.py:12-179 (``minimize``: signature, defaults,
validation, sync rule, seeding), :182-321 (``cpso`` loop: init V=0, pbest=X; per-generation
draw order r1 then r2; return_all; callback; restart after the callback) and
stochopy/optimize/pso/_pso.py:9-122 (PSO = CPSO with competitivity None).  The
per-generation work -- mutation (:324-329), Shrink (cpso/_constraints.py:44-53),
selection_sync and the objective -- is one fused HIP kernel plus the one-workgroup
best/termination kernel; the competitive restart (:405-426) is three small kernels
(csrc/sx_pso.hip).
"""
import ctypes as C

import os

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
    inertia=0.7298,
    cognitivity=1.49618,
    sociability=1.49618,
    competitivity=1.0,
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
    strict_updating=None,
    host_workers=None,
    host_backend=None,
):
    """Minimize an objective function using Competitive PSO on MI355X.

    Parameters are those of the reference (cpso/_cpso.py:12-33) plus ``rng``
    ("numpy-legacy" default = the reference's stream, or "philox" = in-kernel draws);
    ``backend`` must be ``"hip"``.  ``updating="immediate"`` (the reference's default) runs pso_async
    (cpso/_cpso.py:364-402) as one ordered sweep per generation on one GPU whenever that is possible
    (``workers=1``, a factory objective; same seed, same result as the reference's default call); otherwise the
    run is deferred like with a parallel backend of the reference (cpso/_cpso.py:147-150), with a warning.
    ``updating="deferred"`` is the throughput mode; ``strict_updating=False`` forces it silently.
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
    if not 0.0 <= inertia <= 1.0:
        raise ValueError()
    if not 0.0 <= cognitivity <= 4.0:
        raise ValueError()
    if not 0.0 <= sociability <= 4.0:
        raise ValueError()
    if competitivity is not None and not 0.0 <= competitivity <= 2.0:
        raise ValueError()
    if updating not in {"immediate", "deferred"}:
        raise ValueError()
    if constraints not in (None, "Shrink"):
        raise KeyError(constraints)
    if callback is not None and not hasattr(callback, "__call__"):
        raise ValueError()
    _common.resolve_backend(backend, fun_id)
    rng = _common.resolve_rng(rng)
    workers = _common.resolve_workers(workers, fun_id)
    run = _PsoRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(inertia), float(cognitivity),
                  float(sociability), competitivity, constraints, float(xtol), float(ftol), bool(return_all),
                  float(verbosity), callback, rng, seed, workers,
                  immediate=_common.resolve_updating(updating, strict_updating, workers, fun_id, len(lower)))
    return run.result()


class _PsoRun:
    CHECK_EVERY = 32  # philox mode: the host reads the device state every this many generations
    GRAPH_CHUNK = 16  # generations per hipGraph replay (2 or 5 kernel nodes each)

    def __init__(self, fun_id, lower, upper, x0, maxiter, P, w, c1, c2, gamma, constraints, xtol, ftol, return_all,
                 verbosity, callback, rng, seed, workers, autorun=True, immediate=False):
        self.fun_id, self.lower, self.upper = fun_id, lower, upper
        self.external = None if isinstance(fun_id, int) else fun_id  # caller-supplied objective: move -> fun -> select
        self.maxiter, self.P, self.n = maxiter, P, len(lower)
        self.w, self.c1, self.c2, self.gamma, self.constraints = w, c1, c2, gamma, constraints
        self.xtol, self.ftol = xtol, ftol
        self.return_all, self.verbosity, self.callback = return_all, verbosity, callback
        self.rng, self.seed = rng, seed
        self.world = None
        self.Ptotal = P
        self.row0 = 0
        self.immediate = immediate  # pso_async: one sequential sweep per generation (csrc/sx_async.hip)
        import os

        if workers != 1 and rng != "philox":
            workers = _common.replicated_workers("cpso" if gamma else "pso", workers,
                                                 'rng="numpy-legacy" replays ONE host stream in the order of the whole swarm '
                                                 '(rng="philox" shards: draws keyed by the global row)')
        if workers != 1 or os.environ.get("SX_FORCE_SHARDED") == "1":  # env switch: a 1-rank group (tests)
            from ..parallel import require_world

            self.world = require_world(workers)
            if rng != "philox":
                raise ValueError('a sharded run needs rng="philox" (draws keyed by the global row; see parallel.py)')
            self.row0, self.P = self.world.shard(P)  # self.P is the LOCAL swarm from here on
            if gamma and P % self.world.size != 0:
                # (PSO takes any popsize -- blocks of ceil(P / workers) rows, the last rank short; the competitive restart's
                # swarm-wide selection gathers equal [pbestfit | radii] segments per rank)
                raise ValueError(f"cpso with workers={self.world.size}: popsize={P} must be a multiple of workers (pso, de, "
                                 "cmaes and vdcma take any popsize)")
            if immediate:
                raise ValueError("immediate updating is a single-GPU sweep")
        if immediate and self.external is not None:
            raise ValueError("immediate updating evaluates particles one by one inside the sweep kernel: "
                             "only the factory objectives can do that")
        self.x0 = x0
        self.ctx = _device.Context()
        self._graph = None
        self._rccl_graph = None
        self._rccl_graph_note = None
        # sharded swarm: the per-generation best travels by peer writes over xGMI (one one-workgroup kernel,
        # parallel.PeerExchange) when that transport passes its self-test on every rank, else by one all-gather
        self.px, self.exchange, self.exchange_note = None, None, None
        if self.world is not None:
            self.exchange = "rccl"
            if os.environ.get("SX_EXCHANGE", "auto") != "rccl":
                from ..parallel import PeerExchange

                self.px, self.exchange_note = PeerExchange.negotiate(
                    self.ctx, self.world, self.n, float(os.environ.get("SX_XCHG_TIMEOUT_S", "20")))
                if self.px is not None:
                    self.exchange = "p2p"
                elif os.environ.get("SX_EXCHANGE") == "p2p":
                    raise RuntimeError(f"SX_EXCHANGE=p2p is not available: {self.exchange_note}")
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

    def close(self):
        if self._rccl_graph is not None:
            self.ctx.sync()
            self._rccl_graph = None
        if self._graph is not None:
            self.ctx.sync()
            self.ctx.L.sx_graph_destroy(self._graph)
            self._graph = None
        if getattr(self, "_chain_graphs", None):
            self.ctx.sync()
            for g in self._chain_graphs.values():
                self.ctx.L.sx_graph_destroy(g)
            self._chain_graphs = {}
        if self.px is not None:
            self.ctx.sync()
            self.px.close()
            self.px = None

    # ------------------------------------------------------------------ setup
    def _setup(self):
        ctx, P, n = self.ctx, self.P, self.n
        t = _device.torch()
        self.stream = _rng.make_init_stream(self.rng, self.seed)
        if self.gamma:  # cpso/_cpso.py:215-216 (depends on maxiter; the whole swarm's size)
            self.delta = np.log(1.0 + 0.003 * self.Ptotal) / np.max((0.2, np.log(0.01 * self.maxiter)))
        self.d_lower = ctx.upload(self.lower)
        self.d_upper = ctx.upload(self.upper)
        if self.x0 is None and self.rng == "philox":
            # in-kernel draws: the Latin hypercube is drawn on the device too, every rank its own rows (_rng.py)
            self.X = _rng.philox_latin_hypercube(ctx, ctx.empty((P, n)), self.row0, self.Ptotal, self.d_lower,
                                                 self.d_upper, self.seed)
        else:
            if self.x0 is not None:
                X0 = np.array(self.x0, dtype=np.float64)
            else:
                X0 = self.stream.latin_hypercube(self.Ptotal, n, self.lower, self.upper)
            if self.world is not None:
                X0 = np.ascontiguousarray(X0[self.row0 : self.row0 + P])
            self.X = ctx.upload(X0)
        self.V = ctx.zeros((P, n))
        self.pbest = self.X.clone()
        npart = int(ctx.L.sx_num_partials(P, n))
        self.npart = npart
        # [pbestfit | partial radii] in one buffer: with workers > 1 the restart all-gathers exactly this
        self.fit_radius = ctx.empty((P + npart,))
        self.pbestfit = self.fit_radius[:P]
        self.part_r = self.fit_radius[P:]
        if self.world is not None and self.gamma:
            self.fit_radius_all = ctx.empty((self.world.size, P + npart))
        self.candfit = ctx.empty((P,))
        self.part_f = ctx.empty((npart,))
        self.part_i = ctx.empty((npart,), dtype=t.int64)
        self.sel3 = ctx.zeros((3,), dtype=t.int64)
        _common.evaluate_rows(ctx, self.fun_id, self.X, n, self.pbestfit)
        self.candfit.copy_(self.pbestfit)
        out_i = ctx.empty((1,), dtype=t.int64)
        out_f = ctx.empty((1,))
        _lib.check(ctx.L.sx_argmin(_device.ptr(self.pbestfit), P, _device.ptr(self.part_f), _device.ptr(self.part_i),
                                   npart, _device.ptr(out_i), _device.ptr(out_f), ctx.stream_ptr), "sx_argmin")
        g = int(out_i.cpu()[0])
        gfit0 = float(out_f.cpu()[0])
        self.gbest = self.X[g].clone()
        if self.world is not None:  # initial global best: one record exchange, settled on the host
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
        self.state = ctx.upload(np.frombuffer(bytes(st), dtype=np.int64).copy())
        key0, key1 = _rng.philox_key(self.seed) if self.rng == "philox" else (0, 0)
        a = _lib.SxPsoArgs()
        a.X, a.V, a.pbest = self.X.data_ptr(), self.V.data_ptr(), self.pbest.data_ptr()
        a.pbestfit, a.candfit, a.gbest = self.pbestfit.data_ptr(), self.candfit.data_ptr(), self.gbest.data_ptr()
        a.lower, a.upper, a.state = self.d_lower.data_ptr(), self.d_upper.data_ptr(), self.state.data_ptr()
        a.part_f, a.part_i = self.part_f.data_ptr(), self.part_i.data_ptr()
        a.P, a.ld, a.row0, a.n = P, n, self.row0, n
        a.fun_id = self.fun_id if self.external is None else 0
        if self.external is not None:
            self.cand_f = ctx.empty((P,))
        a.constraints = 1 if self.constraints == "Shrink" else 0
        a.rng = _lib.SX_RNG_PHILOX if self.rng == "philox" else _lib.SX_RNG_HOST
        a.maxiter = self.maxiter
        a.w, a.c1, a.c2, a.xtol, a.ftol = self.w, self.c1, self.c2, self.xtol, self.ftol
        a.key0, a.key1 = key0, key1
        self.args = a
        # One kernel per generation (csrc/sx_pso.hip, CHAIN): plain PSO on one GPU with in-kernel draws and nothing to
        # report per generation.  The best / termination step moves into the next launch's prologue; gbest is read from
        # per-workgroup best-row copies (best_rows), the state becomes three words and the records two sets.
        # OPT-IN (SX_PSO_CHAIN=1): bit-identical, but measured SLOWER than the generation + select_finalize pair at
        # BASELINE config 3 -- 37.3 against 35.3 us per generation (profiles/r3_pso_chain.txt): unlike DE's 256
        # workgroups, PSO's 2 048 make 2.7 rounds over the chip and each round pays the prologue's extra dependent trip
        # to memory (records -> best row), which costs what the second kernel did.
        self.chain = bool(self.world is None and self.rng == "philox" and self.external is None and not self.immediate
                          and not self.gamma and self.callback is None and not self.return_all
                          and os.environ.get("SX_PSO_CHAIN", "0") == "1" and ctx.L.sx_pso_chain_supported(C.byref(a)))
        self.launches = 0
        self._chain_graphs = {}
        if self.chain:
            rpb = int(ctx.L.sx_rows_per_workgroup(n))
            # many cheap generations per look at the device (a look is a finalise-only launch + a synchronisation)
            self.CHECK_EVERY, self.GRAPH_CHUNK = 256, 32
            fit = self.pbestfit.cpu().numpy()
            pf = np.full((2, npart), np.inf)
            pi = np.full((2, npart), np.iinfo(np.int64).max, dtype=np.int64)
            rows = np.empty(npart, dtype=np.int64)
            for b in range(npart):
                lo = b * rpb
                k = lo + int(np.argmin(fit[lo : lo + rpb]))  # first minimum, like the kernel's record
                rows[b], pf[0, b], pi[0, b] = k, fit[k], 2 * k
            self.part_f, self.part_i = ctx.upload(pf), ctx.upload(pi)
            self.best_rows = ctx.empty((2, npart, n))
            self.best_rows[0].copy_(self.X[t.from_numpy(rows).to(ctx.device)])
            self.rows_per_block = rpb
            s0 = _lib.SxState(it=0, gbidx=g, gfit=gfit0, dx=0.0, status=_lib.SX_STATUS_NONE, done=0)
            s0.reserved[1] = 2 * g
            st.reserved[0] = st.reserved[1] = 2 * g
            self.state = ctx.upload(np.frombuffer(bytes(s0) + bytes(st) + bytes(st), dtype=np.int64).copy())
            a.state, a.part_f, a.part_i = self.state.data_ptr(), self.part_f.data_ptr(), self.part_i.data_ptr()
            # (a.gbest stays a valid pointer -- the chained kernel never touches it)
        if self.rng == "numpy-legacy":
            self.h_r = [t.empty((P, n), dtype=t.float64).pin_memory() for _ in range(2)]
            self.d_r = [ctx.empty((P, n)) for _ in range(2)]
            a.r1, a.r2 = self.d_r[0].data_ptr(), self.d_r[1].data_ptr()
        if self.world is not None and (self.return_all or self.callback is not None):
            self.Xfull = ctx.empty((self.Ptotal, n))
            self.candfull = ctx.empty((self.Ptotal,))
        if self.return_all:
            self.nout = int(np.ceil(self.verbosity * self.Ptotal))
            rows = max(self.nout, 1)
            self.xall = ctx.empty((self.maxiter, rows, n))
            self.funall = ctx.empty((self.maxiter, rows))
            if self.nout > 0:
                X1, f1 = self._whole_swarm()  # candfit == pbestfit for the initial swarm
                self.xall[0].copy_(X1[: self.nout])
                self.funall[0].copy_(f1[: self.nout])
            else:
                self.xall[0, 0].copy_(self.gbest)
                self.funall[0, 0] = st.gfit
        self.st = st
        self.restarts = []

    # --------------------------------------------------------------- helpers
    def _whole_swarm(self):
        """(positions, their fitness) as the caller sees them: with workers > 1 every rank gathers all shards
        (callbacks / return_all only)."""
        if self.world is None:
            return self.X, self.candfit
        self.world.all_gather_rows(self.X, self.Xfull)
        self.world.all_gather_rows(self.candfit, self.candfull)
        return self.Xfull, self.candfull

    def _record(self, it):
        if not self.return_all:
            return
        X, cand = self._whole_swarm()
        if self.nout > 0:
            self.xall[it - 1].copy_(X[: self.nout])
            self.funall[it - 1].copy_(cand[: self.nout])
        else:
            k = int(cand.argmin())
            self.xall[it - 1, 0].copy_(X[k])
            self.funall[it - 1, 0] = cand[k]

    def _partial_result(self, st):
        res = OptimizeResult(x=self.gbest.cpu().numpy(), fun=st.gfit, nfev=st.it * self.Ptotal, nit=st.it)
        if self.return_all:
            res.update({"xall": self.xall[: st.it].cpu().numpy(), "funall": self.funall[: st.it].cpu().numpy()})
        return res

    def _generation(self):
        ctx = self.ctx
        if self.rng == "numpy-legacy":  # cpso/_cpso.py:262-263: r1 then r2
            for h, d in zip(self.h_r, self.d_r):
                self.stream.random(None, out=h.numpy())
                d.copy_(h, non_blocking=True)
        if self.immediate:
            _lib.check(ctx.L.sx_pso_async_generation(C.byref(self.args), ctx.stream_ptr), "sx_pso_async_generation")
            return
        p, n = _device.ptr, self.n
        if self.external is not None:
            # around the caller's objective (csrc/sx_unfused.hip): move, evaluate the new positions, pbest selection
            _lib.check(ctx.L.sx_pso_move(C.byref(self.args), ctx.stream_ptr), "sx_pso_move")
            self.cand_f.copy_(self.external(ctx, self.X))
            _lib.check(ctx.L.sx_rows_select(p(self.X), n, p(self.cand_f), p(self.pbest), p(self.pbest), n,
                                            p(self.pbestfit), p(self.candfit), self.P, n, p(self.state),
                                            p(self.part_f), p(self.part_i), ctx.stream_ptr), "sx_rows_select")
            if self.world is None:
                _lib.check(ctx.L.sx_select_finalize(p(self.part_f), p(self.part_i), self.npart, p(self.pbest),
                                                    p(self.pbest), n, n, p(self.gbest), p(self.state), self.maxiter,
                                                    self.xtol, self.ftol, ctx.stream_ptr), "sx_select_finalize")
                return
        elif self.world is None:
            _lib.check(ctx.L.sx_pso_generation(C.byref(self.args), 1, ctx.stream_ptr), "sx_pso_generation")
            return
        else:  # sharded swarm: local generation, then the global-best exchange (parallel.py)
            _lib.check(ctx.L.sx_pso_generation(C.byref(self.args), 0, ctx.stream_ptr), "sx_pso_generation")
        if self.px is not None:
            _lib.check(ctx.L.sx_xchg_finalize(p(self.part_f), p(self.part_i), self.npart, p(self.pbest), p(self.pbest),
                                              n, n, self.row0, p(self.gbest), p(self.state), self.maxiter, self.xtol,
                                              self.ftol, C.byref(self.px.args), ctx.stream_ptr), "sx_xchg_finalize")
            return
        _lib.check(ctx.L.sx_shard_best(p(self.part_f), p(self.part_i), self.npart, p(self.pbest), p(self.pbest), n, n,
                                       p(self.state), self.row0, p(self.record), ctx.stream_ptr), "sx_shard_best")
        self.world.all_gather_records(self.record, self.records)
        _lib.check(ctx.L.sx_gather_finalize(p(self.records), self.world.size, n, p(self.gbest), p(self.state),
                                            self.maxiter, self.xtol, self.ftol, ctx.stream_ptr), "sx_gather_finalize")

    def _restart_device(self):
        """cpso/_cpso.py:405-426 entirely on the device (Philox positions keyed by row)."""
        ctx, a = self.ctx, C.byref(self.args)
        _lib.check(ctx.L.sx_pso_radius(a, _device.ptr(self.part_r), ctx.stream_ptr), "sx_pso_radius")
        if self.world is None:
            _lib.check(ctx.L.sx_pso_restart_select(a, _device.ptr(self.part_r), float(self.delta), float(self.gamma),
                                                   _device.ptr(self.sel3), ctx.stream_ptr), "sx_pso_restart_select")
        else:
            # the swarm radius is a max and the worst-nw rule a rank over ALL particles: one all-gather of
            # [pbestfit | partial radii] per generation, then every rank derives the same threshold
            self.world.all_gather_records(self.fit_radius, self.fit_radius_all)
            _lib.check(ctx.L.sx_pso_restart_select_gathered(a, _device.ptr(self.fit_radius_all), self.world.size,
                                                            float(self.delta), float(self.gamma),
                                                            _device.ptr(self.sel3), ctx.stream_ptr),
                       "sx_pso_restart_select_gathered")
        _lib.check(ctx.L.sx_pso_restart_apply(a, _device.ptr(self.sel3), None, None, 0, ctx.stream_ptr),
                   "sx_pso_restart_apply")

    def _restart_host_order(self, it):
        """numpy-legacy stream: the new positions are drawn on the host for the rows in the reference's
        descending-fitness order (cpso/_cpso.py:420-422), so row selection happens on the host too."""
        ctx, P, n = self.ctx, self.P, self.n
        _lib.check(ctx.L.sx_pso_radius(C.byref(self.args), _device.ptr(self.part_r), ctx.stream_ptr), "sx_pso_radius")
        radius = float(self.part_r.max().cpu()) / np.sqrt(4.0 * n)
        if not radius < self.delta:
            return
        inorm = it / self.maxiter
        nw = int((P - 1.0) / (1.0 + np.exp(1.0 / 0.09 * (inorm - self.gamma + 0.5))))
        if nw <= 0:
            return
        rows = self.pbestfit.cpu().numpy().argsort()[: -nw - 1 : -1]
        newx = self.stream.uniform_rows(self.lower, self.upper, nw)
        d_rows = ctx.upload(np.ascontiguousarray(rows, dtype=np.int64))
        d_newx = ctx.upload(newx)
        _lib.check(ctx.L.sx_pso_restart_apply(C.byref(self.args), None, _device.ptr(d_rows), _device.ptr(d_newx), nw,
                                              ctx.stream_ptr), "sx_pso_restart_apply")
        ctx.sync()  # d_rows / d_newx must outlive the kernel
        self.restarts.append((it, nw))

    # ------------------------------------------------------------------ loop
    def _run(self):
        ctx = self.ctx
        self._setup()
        st = self.st
        if self.callback is not None:
            self.callback(self._whole_swarm()[0].cpu().numpy(), self._partial_result(st))
        # return_all with in-kernel draws: history copies (cpso/_cpso.py:283-295) are device-side and ordered on
        # the engine stream, so the host need not look at every generation
        record_async = (self.return_all and self.rng == "philox" and self.callback is None and self.nout > 0
                        and self.maxiter > 1)
        stepwise = (self.rng == "numpy-legacy" or self.callback is not None or self.return_all) and not record_async
        while not st.done:
            if record_async:
                for j in range(min(max(self.maxiter - st.it, 1), self.CHECK_EVERY)):
                    self._generation()
                    self._record(st.it + 1 + j)  # generations after convergence are no-ops; their slots are cut off
                    if self.gamma:
                        self._restart_device()
                st = ctx.read_state(self.state)
            elif stepwise:
                self._generation()
                self._record(st.it + 1)
                st = ctx.read_state(self.state)
                if self.callback is not None:
                    self.callback(self._whole_swarm()[0].cpu().numpy(), self._partial_result(st))
                if not st.done and self.gamma:
                    if self.rng == "numpy-legacy":
                        self._restart_host_order(st.it)
                    else:
                        self._restart_device()
            elif self.immediate or self.external is not None:
                # long sweeps: look after every few of them.  A caller's device objective between our kernels:
                # chunks of generations captured into one graph (kernels + objective) and replayed, else eagerly
                look = 8 if self.immediate else self.CHECK_EVERY
                todo = min(max(self.maxiter - st.it, 1), look)
                while self.external is not None and todo >= self.GRAPH_CHUNK and self._capture_sharded_chunk():
                    self._rccl_graph.replay()
                    todo -= self.GRAPH_CHUNK
                for _ in range(todo):
                    self._generation()
                    if self.gamma:
                        self._restart_device()
                st = ctx.read_state(self.state)
            else:
                self.enqueue(min(max(self.maxiter - st.it, 1), self.CHECK_EVERY))
                st = self.read_state()
                if self.px is not None and self.px.failed():
                    raise RuntimeError("peer exchange timed out: a rank did not reach the generation the others "
                                       "were waiting for (SX_XCHG_TIMEOUT_S)")
        self.st = st
        status = int(st.status)
        xbest = self.gbest.cpu().numpy()
        if self.chain:
            xbest = self._chain_row(st.reserved[1])
            # the chained kernel stops on `fun <= ftol` with status 1; _common.py:135-140 calls it 0 when the best moved
            # by <= xtol.  Both generations' best rows are still resident (nothing is produced after `done`).
            if status == 1 and np.linalg.norm(self._chain_row(st.reserved[0]) - xbest) <= self.xtol:
                status = 0
        res = OptimizeResult(
            x=xbest,
            success=status >= 0,
            status=status,
            message=_common.messages[status],
            fun=float(st.gfit),
            nfev=int(st.it) * self.Ptotal,
            nit=int(st.it),
        )
        if self.return_all:
            res.update({"xall": self.xall[: st.it].cpu().numpy(), "funall": self.funall[: st.it].cpu().numpy()})
        # pso_async assigns X[i] row by row, i.e. works in place on the caller's x0 (cpso/_cpso.py:389)
        if self.immediate and isinstance(self.x0, np.ndarray) and self.x0.dtype == np.float64:
            self.x0[...] = self.X.cpu().numpy()
        if self.rng == "numpy-legacy":
            self.stream.sync_back()
        ctx.sync()
        # (peer exchange: the one meeting point of all ranks -- success flag included -- is `all_agree` in the caller's
        #  `finally`; a barrier here would pair with a failing rank's all_gather there: mismatched collectives)
        self._res = res

    # ---- chained mode (one kernel per generation) ----
    def _chain_launch(self, parity, finalize_only):
        _lib.check(self.ctx.L.sx_pso_chain_launch(C.byref(self.args), _device.ptr(self.best_rows), parity, finalize_only,
                                                  self.ctx.stream_ptr), "sx_pso_chain_launch")

    def _chain_graph(self, par, size):
        key = (par, size)
        if key not in self._chain_graphs:
            g = C.c_void_p()
            _lib.check(self.ctx.L.sx_pso_chain_graph_create(C.byref(self.args), _device.ptr(self.best_rows), size, par,
                                                            C.byref(g)), "sx_pso_chain_graph_create")
            self._chain_graphs[key] = g
        return self._chain_graphs[key]

    def _enqueue_chain(self, ngen):
        while ngen >= self.GRAPH_CHUNK:  # (GRAPH_CHUNK is even: a replay leaves the launch parity where it found it)
            _lib.check(self.ctx.L.sx_graph_launch(self._chain_graph(self.launches & 1, self.GRAPH_CHUNK),
                                                  self.ctx.stream_ptr), "sx_graph_launch")
            self.launches += self.GRAPH_CHUNK
            ngen -= self.GRAPH_CHUNK
        for _ in range(ngen):
            self._chain_launch(self.launches & 1, 0)
            self.launches += 1

    def read_state(self):
        """Host view of the run: (chained mode) finalise the last generation into state[2], then read it."""
        if not self.chain:
            return self.ctx.read_state(self.state)
        self._chain_launch(self.launches & 1, 1)
        return self.ctx.read_state(self.state[16:24])

    def _chain_row(self, rec):
        """The best row a record points to: best_rows[q][workgroup of the row]."""
        rec = int(rec)
        return self.best_rows[rec & 1, (rec >> 1) // self.rows_per_block].cpu().numpy()

    def enqueue(self, ngen):
        """Enqueue `ngen` generations (and their restarts) without host synchronisation (Philox mode).
        Single GPU: full chunks replay one instantiated hipGraph of the loop body."""
        ctx = self.ctx
        if self.chain:
            self._enqueue_chain(ngen)
            return
        if self.world is None:
            while ngen >= self.GRAPH_CHUNK:
                if self._graph is None:
                    g = C.c_void_p()
                    restart = bool(self.gamma)
                    _lib.check(ctx.L.sx_pso_graph_create(
                        C.byref(self.args), self.GRAPH_CHUNK, _device.ptr(self.part_r) if restart else None,
                        float(self.delta) if restart else 0.0, float(self.gamma) if restart else 0.0,
                        _device.ptr(self.sel3) if restart else None, C.byref(g)), "sx_pso_graph_create")
                    self._graph = g
                _lib.check(ctx.L.sx_graph_launch(self._graph, ctx.stream_ptr), "sx_graph_launch")
                ngen -= self.GRAPH_CHUNK
        elif self.world is not None:
            # sharded swarm: kernels + the RCCL all-gathers of a chunk of generations captured once and replayed
            while ngen >= self.GRAPH_CHUNK and self._capture_sharded_chunk():
                self._rccl_graph.replay()
                ngen -= self.GRAPH_CHUNK
        for _ in range(ngen):
            self._generation()
            if self.gamma:
                self._restart_device()

    def _capture_sharded_chunk(self):
        """GRAPH_CHUNK sharded generations (kernels + all-gathers) as one graph; False if that is not possible
        (gloo stages through the host; SX_RCCL_GRAPH=0; a failed capture) -- same collectives either way."""
        import os

        if self._rccl_graph is not None:
            return True
        if (self._rccl_graph_note is not None or (self.world is not None and self.world.backend != "nccl")
                or os.environ.get("SX_RCCL_GRAPH") == "0"
                or (self.external is not None and (os.environ.get("SX_EXT_GRAPH") == "0"
                                                   or not getattr(self.external, "capturable", True)))):
            return False
        t = _device.torch()
        try:
            self.ctx.sync()
            if self.world is not None:
                self.world.quiesce_for_capture(self.ctx)
            g = t.cuda.CUDAGraph()
            with t.cuda.graph(g, stream=self.ctx.stream, capture_error_mode=_CAPTURE_MODE):
                for _ in range(self.GRAPH_CHUNK):
                    self._generation()
                    if self.gamma:
                        self._restart_device()
            self._rccl_graph = g
            return True
        except Exception as e:  # capture is an optimisation, never a requirement
            self._rccl_graph_note = f"graph capture of the rccl path failed: {e}"
            return False

    def result(self):
        return self._res


register("cpso", minimize)
