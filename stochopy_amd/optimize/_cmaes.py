"""CMA-ES front end + generation loop for ``backend="hip"``.

Reference: stochopy/optimize/cmaes/_cmaes.py:12-140 (``minimize``) and :143-357 (``cmaes``
loop), :360-434 (``converge``).  On the device (csrc/sx_cmaes.hip, csrc/sx_core.hip):
sampling ``arx = xmean + sigma*B*(D o z)`` and the covariance update as fp64 MFMA
contractions, the objective (fused un-standardisation), the recombination of the
mean.  On the host, as in the reference: ranking (``np.argsort``), the evolution paths,
the step size, the ten stopping rules and -- SURVEY.md section 8f rank 1, "next" -- the
eigendecomposition (``numpy.linalg.eigh`` = LAPACK, the reference's own third-party call),
which also fixes the eigenvector signs the same-seed parity depends on.
``constraints="Penalize"`` (cmaes/_constraints.py:4-82): the clipping of the
candidates, the objective and the weighted squared excess run in one device kernel
(``sx_cmaes_eval_penalized``); the scalar bookkeeping of the boundary weights stays on the host.
"""

import os

import numpy as np

from .. import _device, _lib, _rng
from ..linalg import Eigh
from . import _common
from ._helpers import OptimizeResult, register

__all__ = ["minimize"]


def minimize(
    fun,
    bounds,
    x0=None,
    args=(),
    maxiter=100,
    popsize=10,
    sigma=0.1,
    muperc=0.5,
    seed=None,
    xtol=1.0e-8,
    ftol=1.0e-8,
    constraints=None,
    workers=1,
    backend=None,
    return_all=False,
    verbosity=1.0,
    callback=None,
    rng=None,
    eigh=None,
    host_workers=None,
    host_backend=None,
):
    """Minimize an objective function using CMA-ES on MI355X (reference cmaes/_cmaes.py:12-30).

    ``eigh="host"`` decomposes C with numpy/LAPACK exactly like the reference (cmaes/_cmaes.py:304 -- the
    reference's own third-party call), whose eigenvector SIGNS (an accident of LAPACK's internals) same-seed
    parity with the reference depends on: the default in the parity mode ``rng="numpy-legacy"``.
    ``eigh="device"`` keeps C on the GPU and decomposes it with this package's own eigensolver
    (csrc/sx_eigh.hip: parallel block Jacobi on the fp64 matrix cores): same
    eigenvalues and eigenvectors to rounding, signs by a stated rule (largest component positive), no
    2 x n^2 PCIe trip -- the default in the throughput mode ``rng="philox"``; the oracle reproduces such runs
    with ``eigh="canonical"`` (LAPACK + the same sign rule).
    ``eigh=callable``: ``callable(C) -> (eigenvalues, eigenvectors)`` replaces the host decomposition (another solver,
    or -- tests/test_gpu_cmaes.py -- the reference's own eigenpairs replayed, which makes same-seed parity checkable
    for shapes whose covariance has a repeated eigenvalue, where no two solvers agree on a basis).

    ``workers > 1`` (one process per GPU) shards what the reference's parallel backends shard -- the
    candidates: every rank samples and evaluates ``popsize / workers`` rows (same draws: the legacy stream is
    replicated on the hosts, Philox normals are keyed by the global row), one all-gather per generation
    returns all candidates and fitness values to every rank, and the O(n^2)/O(n^3) model update is replicated.
    Same result as ``workers=1`` on every rank.
    """
    fun_id = _common.resolve_objective(fun, args, workers, backend, host_workers, host_backend)
    lower, upper = _common.as_bounds(bounds)
    if x0 is not None:
        if np.ndim(x0) != 1 or len(x0) != len(bounds):
            raise ValueError()
    if sigma <= 0.0:
        raise ValueError()
    if not 0.0 < muperc <= 1.0:
        raise ValueError()
    if constraints not in (None, "Penalize"):
        raise KeyError(constraints)
    if callback is not None and not hasattr(callback, "__call__"):
        raise ValueError()
    _common.resolve_backend(backend, fun_id)
    rng = _common.resolve_rng(rng)
    workers = _common.resolve_workers(workers, fun_id)
    if len(lower) > _lib.NARROW_DIM:
        # the one method that keeps a dimension cap: an n x n covariance, its eigenvectors and an O(n^3) decomposition per
        # update (cmaes/_cmaes.py:290-309).  Every other method serves rows of up to _lib.WIDE_DIM elements.
        raise ValueError(f"method='cmaes' keeps the full {len(lower)} x {len(lower)} covariance and its eigendecomposition on "
                         f"the device: ndim <= {_lib.NARROW_DIM}.  method='vdcma' is the O(n) variant for long vectors.")
    if eigh is None:
        eigh = "host" if rng == "numpy-legacy" else "device"
    if not callable(eigh) and eigh not in ("host", "device"):
        raise ValueError("eigh must be 'host', 'device' or a callable C -> (eigenvalues, eigenvectors)")
    if (rng == "philox" and eigh == "device" and isinstance(fun_id, int) and os.environ.get("SX_CMA_LOOP", "") != "host"
            and (constraints is None or 20.0 + 3.0 * len(lower) / int(popsize) + 1.0 <= 256.0)):
        # nothing the host has to COMPUTE between generations: the whole loop (and the history) stays on the device --
        # since round 3 including constraints="Penalize" (its boundary-weight bookkeeping: cma_penalty_kernel),
        # workers > 1 (every rank samples and evaluates its share of the candidates, one all-gather, replicated model)
        # and a callback (the host then looks at every generation and copies what the callback is shown).
        # SX_CMA_LOOP=host: the host-driven loop with the device eigensolver (tests, tools/bench_c4.py).
        return _CmaDeviceRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(sigma), float(muperc),
                             float(xtol), float(ftol), seed, bool(return_all), float(verbosity),
                             penalize=constraints == "Penalize", workers=workers, callback=callback).result()
    run = _CmaRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(sigma), float(muperc), float(xtol),
                  float(ftol), bool(return_all), float(verbosity), callback, rng, seed, eigh, workers,
                  penalize=constraints == "Penalize")
    return run.result()


def _strategy_constants(n, P, muperc):
    """cmaes/_cmaes.py:184-205."""
    mu = int(muperc * P)
    w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    w /= w.sum()
    mueff = w.sum() ** 2 / np.square(w).sum()
    cc = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)
    cs = (mueff + 2.0) / (n + mueff + 5.0)
    c1 = 2.0 / ((n + 1.3) ** 2 + mueff)
    cmu = min(1.0 - c1, 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))
    damps = 1.0 + 2.0 * max(0.0, np.sqrt((mueff - 1.0) / (n + 1.0)) - 1.0) + cs
    chind = np.sqrt(n) * (1.0 - 1.0 / (4.0 * n) + 1.0 / (21.0 * n**2))
    return mu, w, mueff, cc, cs, c1, cmu, damps, chind


class _CmaDeviceRun:
    """The reference's loop (cmaes/_cmaes.py:226-343) with every per-generation step on the device
    (csrc/sx_cma_loop.hip): the host enqueues generations -- it only works out, from the generation number, when the
    eigendecomposition is due (:301) -- and looks at the 128-byte state every LOOK generations (every generation
    while one takes milliseconds).  Draws: Philox normals keyed by (seed, generation, row); eigenvectors with the
    canonical sign of csrc/sx_eigh.hip."""

    LOOK = 16
    COLD_SWEEPS = 40   # launched for a decomposition started from the identity (n=512: 11-13 are carried out)
    WARM_SWEEPS0 = 16  # launched for the first warm-started one; afterwards: what the last one needed + 1

    def __init__(self, fun_id, lower, upper, x0, maxiter, P, sigma, muperc, xtol, ftol, seed, return_all=False,
                 verbosity=1.0, run=True, penalize=False, workers=1, callback=None):
        """``run=False`` only builds the device state (``self.buffers``, ``self.args``): tests drive single generations
        with ``step`` from a state of their choosing."""
        import ctypes as C
        import warnings

        ctx = self.ctx = _device.Context()
        t = _device.torch()
        L, ptr, n = ctx.L, _device.ptr, len(lower)
        world, row0, Pl = None, 0, P
        if workers != 1:
            from ..parallel import require_world

            world = require_world(workers)
            row0, Pl = world.shard(P)  # blocks of ceil(P / workers) rows, the last rank short
        with t.cuda.stream(ctx.stream):
            init = _rng.make_init_stream("philox", seed)
            key0, key1 = _rng.philox_key(seed)
            xm, xstd = 0.5 * (upper + lower), 0.5 * (upper - lower)
            xmean = init.uniform(-1.0, 1.0, n) if x0 is None else (np.asarray(x0, dtype=np.float64) - xm) / xstd
            mu, w, mueff, cc, cs, c1, cmu, damps, chind = _strategy_constants(n, P, muperc)
            eig = Eigh(ctx, n)
            keep = self._keep = dict(
                Z=ctx.empty((P, n)), arx=ctx.empty((P, n)), fit=ctx.empty((P,)), xmean=ctx.upload(xmean),
                xold=ctx.zeros((n,)), ps=ctx.zeros((n,)), pc=ctx.zeros((n,)), C=ctx.upload(np.eye(n)),
                B=ctx.upload(np.eye(n)), D=ctx.upload(np.ones(n)), eigw=eig.w, w=ctx.upload(w), Y=ctx.empty((mu, n)),
                part=ctx.empty((64, n)), step=ctx.empty((n,)), isc=ctx.empty((n,)), xnew=ctx.empty((n,)),
                ypart=ctx.empty((8, n)), besthist=ctx.zeros((maxiter,)), xm=ctx.upload(xm), xstd=ctx.upload(xstd),
                xbest=ctx.zeros((n,)), order=ctx.empty((P,), dtype=t.int64), eigh_ws=eig.ws)
            if penalize:  # cmaes/_constraints.py:4-82 on the device: weights 0, spread history [1.0], both phase flags as at :213-215
                pw = np.zeros(2 * n + P + 256 + 4)
                pw[2 * n + P] = 1.0
                pw[2 * n + P + 256: 2 * n + P + 259] = (1.0, 0.0, 1.0)
                keep["pen_ws"] = ctx.upload(pw)
                keep["pen_order"] = ctx.empty((P,), dtype=t.int64)
            nout = int(np.ceil(verbosity * P)) if return_all else 0
            if return_all:  # device-side history slabs, read back once at the end
                keep["hist_x"] = ctx.empty((maxiter, max(1, nout), n))
                keep["hist_f"] = ctx.empty((maxiter, max(1, nout)))
            st = _lib.SxCmaState(it=0, nfev=0, best_row=0, fbest=0.0, sigma=sigma, sigma_next=sigma, tmp_coef=0.0,
                                 psnorm=0.0, status=_lib.SX_STATUS_NONE, done=0, stop_it=0)
            d_state = keep["state"] = ctx.upload(np.frombuffer(bytes(st), dtype=np.float64))
            a = _lib.SxCmaArgs(**{k: ptr(v) for k, v in keep.items()})
            a.eigh_ws_bytes, a.P, a.n, a.mu, a.fun_id, a.maxiter = eig.bytes, P, n, mu, fun_id, maxiter
            a.hist_rows = nout
            a.ilim, a.eig_sweeps = int(10.0 + 30.0 * n / P), self.COLD_SWEEPS
            a.cs, a.cc, a.c1, a.cmu, a.damps, a.chind, a.mueff = cs, cc, c1, cmu, damps, chind, mueff
            a.xtol, a.ftol, a.insigma, a.key0, a.key1 = xtol, ftol, sigma, key0, key1
            self.buffers, self.args, self.eig, self.P = keep, a, eig, P
            if world is not None:
                arx_loc, fit_loc = ctx.empty((Pl, n)), ctx.empty((Pl,))
            self.eig_every = eig_every = P / (c1 + cmu) / n / 10.0  # :301
            if not run:
                return
            # How often the host looks is a matter of speed only (generations after the stop are no-ops) -- except that the
            # sweep allowance below follows what the host has SEEN.  Both are therefore functions of the problem, not of
            # the clock or of the number of ranks: look_cap generations between looks (about 2 ms of them),
            # reached by doubling, and the same allowance rule whether this run looks that rarely or (callback, workers > 1)
            # at every generation.
            look_cap = 1 if n > 256 else min(self.LOOK, max(1, 512 // max(n, 32)))
            eigeneval, look, since = 0, 1, 0
            cb_hist, cb_pin, fails_seen, warned_short = None, None, 0, False
            pin_state = pin_eig = None
            state = st
            # Sweeps to LAUNCH per decomposition (launches beyond convergence are no-ops of ~2 us each, 2n/16 - 1 per
            # sweep): a cold start gets the full allowance; a warm start what the last one needed + 1 -- inside a run
            # the count moves by at most one between decompositions (tools/eigh_c4_sweeps.py).
            warm_sweeps, launched, decomposed = self.WARM_SWEEPS0, 0, False
            # Large n, one GPU, no callback (the host looks at every generation anyway, look_cap == 1): the decomposition is
            # enqueued in PIECES (sx_cmaes_generation_phased) -- the rounds the last one needed, one look at the solver's run
            # record, then either its finish or another sweep -- instead of a whole sweep of no-op launches "in case" (~30 of
            # 3.5 us per decomposition at n = 512).  That one look per generation also returns the state record the previous
            # generation's stop rules wrote, so the host still waits once per generation (the stop is seen one generation
            # late: what was enqueued in between does nothing).
            rps = int(L.sx_eigh_rounds_per_sweep(n))
            # Round 6: the solver's rounds run inside ONE resident launch that ends itself (sx_eigh_set_flow, csrc/sx_eigh.hip
            # eigh_flow_kernel): nothing is launched "in case", so a decomposition gets the full allowance in one piece and
            # the pieces below are not needed.  Its grid must be on the chip at once: ranks that share a GPU (only possible
            # without RCCL, i.e. the tests' gloo groups) take the launch-per-round form.
            if world is not None and world.backend != "nccl":
                L.sx_eigh_set_flow(0)
            flow = rps > 0 and int(L.sx_eigh_set_flow(-2)) == 1
            phased = (world is None and callback is None and look_cap == 1 and rps > 0 and not flow
                      and os.environ.get("SX_CMA_PHASED", "1") != "0")
            warm_rounds = None  # rounds the last phased decomposition needed
            cap_hit = False     # the last phased decomposition was still open when the round cap was reached
            for gen in range(1, maxiter + 1):
                due = gen * P - eigeneval > eig_every
                if due:  # 1: first decomposition; 2: start from the previous eigenvectors (C changes by O(c1 + cmu))
                    due = 2 if eigeneval else 1
                    eigeneval = gen * P
                    a.eig_sweeps = launched = 60 if flow else (self.COLD_SWEEPS if due == 1 else warm_sweeps)
                    decomposed = True
                if phased and due:
                    if pin_state is None:
                        pin_state = t.empty_like(d_state, device="cpu").pin_memory()
                        pin_eig = t.empty((256,), dtype=t.float64).pin_memory()
                    cap = 60 * rps
                    r1 = min(cap, warm_rounds if (due == 2 and warm_rounds) else 8 * rps + 2)
                    _lib.check(L.sx_cmaes_generation_phased(C.byref(a), gen, int(due), 0, 0, r1, ctx.stream_ptr),
                               "sx_cmaes_generation_phased")
                    while True:
                        pin_state.copy_(d_state, non_blocking=True)
                        pin_eig.copy_(eig.ws[:256], non_blocking=True)
                        ctx.sync()
                        state = _lib.SxCmaState.from_buffer_copy(pin_state.numpy().tobytes())  # (of generation gen - 1)
                        rec = pin_eig.numpy().view(np.int32)  # EighInfo: done_seq, sweeps, parity, converged, ...
                        if state.done or rec[0] != 0 or r1 >= cap:
                            break
                        r0, r1 = r1, min(cap, r1 + rps)
                        _lib.check(L.sx_cmaes_generation_phased(C.byref(a), gen, int(due), 1, r0, r1, ctx.stream_ptr),
                                   "sx_cmaes_generation_phased")
                    if state.done:
                        break
                    # (the look above also shows what the PREVIOUS generation's closing kernel made of a run that had hit the cap)
                    fails = int(rec[510])  # (byte 2040) decompositions since the start that fell short of the tolerance
                    if fails != fails_seen:
                        if cap_hit:
                            warnings.warn("stochopy_amd: the device eigensolver did not reach its tolerance in 60 sweeps; the "
                                          "decomposition is used as it is", RuntimeWarning, stacklevel=3)
                        fails_seen = fails
                    cap_hit = False
                    _lib.check(L.sx_cmaes_generation_phased(C.byref(a), gen, int(due), 2, 0, r1, ctx.stream_ptr),
                               "sx_cmaes_generation_phased")
                    if rec[0] != 0:  # ended within what was enqueued: it carried out rec[1] sweeps and the two rounds that tell
                        warm_rounds = int(rec[1]) * rps + 2
                    else:
                        # the round cap was hit with the run still open.  Phase 2's closing kernel may yet accept the
                        # result through the refinement criterion (eigh_close_kernel) and counts a short-fall on the device
                        # otherwise: the verdict is read from that counter at the next look, not guessed here
                        cap_hit = True
                        warm_rounds = None
                    decomposed, since = False, 0
                    continue
                if world is None:
                    _lib.check(L.sx_cmaes_generation(C.byref(a), gen, int(due), ctx.stream_ptr), "sx_cmaes_generation")
                else:  # own candidates, one gather of candidates and fitness, the model update replicated on every rank
                    _lib.check(L.sx_cmaes_generation_stage(C.byref(a), gen, int(due), 0, row0, Pl, ptr(arx_loc), ptr(fit_loc),
                                                           ctx.stream_ptr), "sx_cmaes_generation_stage")
                    world.all_gather_rows(arx_loc, keep["arx"])
                    world.all_gather_rows(fit_loc, keep["fit"])
                    _lib.check(L.sx_cmaes_generation_stage(C.byref(a), gen, int(due), 1, 0, 0, None, None, ctx.stream_ptr),
                               "sx_cmaes_generation_stage")
                    look = 1  # (every rank must stop enqueueing collectives at the same generation)
                since += 1
                if callback is not None:
                    look = 1  # the callback sees every generation (cmaes/_cmaes.py:333-343)
                if since >= look or gen == maxiter:
                    # ONE trip to the device per look: the state record and (when a decomposition was enqueued since the
                    # last look) the head of the eigensolver's workspace -- its run record and the count of runs that fell
                    # short -- through pinned memory behind a single synchronisation (three blocking copies before)
                    if pin_state is None:
                        pin_state = t.empty_like(d_state, device="cpu").pin_memory()
                        pin_eig = t.empty((256,), dtype=t.float64).pin_memory()
                    pin_state.copy_(d_state, non_blocking=True)
                    if decomposed:
                        pin_eig.copy_(eig.ws[:256], non_blocking=True)
                    ctx.sync()
                    state = _lib.SxCmaState.from_buffer_copy(pin_state.numpy().tobytes())
                    if callback is not None:
                        # what the reference hands over: all candidates (the clipped ones with Penalize), un-standardised,
                        # and the best of them; with return_all the history so far (copied slab by slab)
                        if cb_pin is None:
                            cb_pin = t.empty((P, n), dtype=t.float64).pin_memory()
                        cb_pin.copy_(keep["arx"])
                        rows = cb_pin.numpy()
                        Xs = np.multiply(np.clip(rows, -1.0, 1.0) if penalize else rows, xstd)  # (a new array every generation)
                        Xs += xm
                        cres = OptimizeResult(x=Xs[int(state.best_row)].copy(), fun=float(state.fbest), nfev=gen * P, nit=gen)
                        if return_all:
                            if cb_hist is None:
                                cb_hist = (np.empty(tuple(keep["hist_x"].shape)), np.empty(tuple(keep["hist_f"].shape)))
                            cb_hist[0][gen - 1] = keep["hist_x"][gen - 1].cpu().numpy()
                            cb_hist[1][gen - 1] = keep["hist_f"][gen - 1].cpu().numpy()
                            cres.update({"xall": cb_hist[0][:gen], "funall": cb_hist[1][:gen]})
                        callback(Xs, cres)
                    if state.done:
                        break
                    if world is None and callback is None and look < look_cap:
                        look *= 2  # cheap generations: look less often
                    if decomposed:  # (the record is only meaningful once a decomposition has been enqueued)
                        rec = pin_eig.numpy().view(np.int32)  # csrc/sx_eigh.hip EighInfo: done_seq, sweeps, parity, converged, ...
                        used, ok = int(rec[1]), bool(rec[3])
                        fails = int(rec[510])  # (byte 2040) runs since the start that fell short
                        if ok and fails == fails_seen:
                            # the record is the LAST decomposition's: while the host looks at every generation the count
                            # moves by at most one between looks; between rarer looks it gets two more sweeps of slack
                            warm_sweeps = min(60, used + (1 if look_cap == 1 else 3))
                        else:
                            if launched >= 60:
                                warnings.warn("stochopy_amd: the device eigensolver did not reach its tolerance in 60 "
                                              "sweeps; the decomposition is used as it is", RuntimeWarning, stacklevel=3)
                            elif not warned_short:
                                warnings.warn("stochopy_amd: a warm-started decomposition ran out of its %d launched sweeps "
                                              "short of the tolerance (its result is used as it is); later ones get the "
                                              "full allowance" % launched, RuntimeWarning, stacklevel=3)
                                warned_short = True
                            warm_sweeps, fails_seen = 60, fails
                        decomposed = False
                    since = 0
            if not state.done:  # (the phased loop sees a generation's record one look late)
                state = _lib.SxCmaState.from_buffer_copy(d_state.cpu().numpy().tobytes())
            if cap_hit and int(eig.ws[:256].cpu().numpy().view(np.int32)[510]) != fails_seen:  # the run's last decomposition
                warnings.warn("stochopy_amd: the device eigensolver did not reach its tolerance in 60 sweeps; the "
                              "decomposition is used as it is", RuntimeWarning, stacklevel=3)
            if not state.done:  # cannot happen: generation maxiter sets status -1
                raise RuntimeError("CMA-ES device loop ended without a status")
            nit = int(state.stop_it)
            self._res = OptimizeResult(x=keep["xbest"].cpu().numpy(), success=state.status >= 0, status=int(state.status),
                                       message=_common.messages[int(state.status)], fun=float(state.fbest),
                                       nfev=nit * P, nit=nit)
            if return_all:
                self._res.update({"xall": keep["hist_x"][:nit].cpu().numpy(), "funall": keep["hist_f"][:nit].cpu().numpy()})
            ctx.sync()

    def step(self, gen, due):
        """Enqueue generation ``gen`` (``due``: 0 no decomposition, 1 cold, 2 started from the current B)."""
        import ctypes as C

        t = _device.torch()
        with t.cuda.stream(self.ctx.stream):
            _lib.check(self.ctx.L.sx_cmaes_generation(C.byref(self.args), int(gen), int(due), self.ctx.stream_ptr),
                       "sx_cmaes_generation")

    def read_state(self):
        return _lib.SxCmaState.from_buffer_copy(self.buffers["state"].cpu().numpy().tobytes())

    def result(self):
        return self._res


class _BoundaryWeights:
    """Host bookkeeping of constraints="Penalize" (cmaes/_constraints.py:33-76; state created at
    cmaes/_cmaes.py:213-215, 230-231): per-dimension penalty weights, the sliding history of fitness-spread
    estimates they are initialised from, and the two phase flags.  ``update`` takes the RAW fitness of the
    clipped candidates and returns ``weights / scale`` -- the vector the device multiplies the squared excess by."""

    def __init__(self, n):
        self.weights = np.zeros(n)
        self.spreads = np.ones(1)
        self.have_spread = False
        self.initial_phase = True

    def update(self, fit_raw, xmean, xold, sigma, diagC, mueff, it, P):
        n = xmean.size
        q25, q75 = np.percentile(fit_raw, [25.0, 75.0])
        spread = (q75 - q25) / n / diagC.mean() / sigma**2
        if spread == 0:
            spread = self.spreads[self.spreads > 0.0].min()
        elif not self.have_spread:
            self.spreads = np.empty(0)
            self.have_spread = True
        keep = self.spreads if self.spreads.size < 20 + (3.0 * n) / P else self.spreads[1:]
        self.spreads = np.append(keep, spread)
        outside = (xmean < -1.0) | (xmean > 1.0)
        if outside.any():
            if self.initial_phase:
                self.weights = np.full(n, 2.0002 * np.median(self.spreads))
                if self.have_spread and it > 2:
                    self.initial_phase = False
            # the reference measures the excess against a mean clipped on the UPPER side only (:52-53: the
            # second np.where starts again from xmean), so weights only ever grow for dimensions above +1
            excess = xmean - np.where(xmean > 1.0, 1.0, xmean)
            limit = 3.0 * max(1.0, np.sqrt(n / mueff)) * sigma * np.sqrt(diagC)
            grow = outside & (np.abs(excess) > limit) & (np.sign(excess) == np.sign(xmean - xold))
            self.weights = np.where(grow, self.weights * 1.2 ** min(1.0, mueff / 10.0 / n), self.weights)
        logd = np.log(diagC)
        return self.weights / np.exp(0.9 * (logd - logd.mean()))


def _stop_status(it, n, maxiter, xmean, xold, besthist, arfit, order, sigma, insigma, ilim, pc, xtol, ftol, diagC, B, D):
    """The ten ordered stopping rules of cmaes/_cmaes.py:360-434 (including the zero-padded history reads)."""
    axis = int(np.floor(np.mod(it, n)))
    sd = np.sqrt(diagC)
    fbest = arfit[order[0]]
    if it >= maxiter:
        return -1
    if np.linalg.norm(xold - xmean) <= xtol and fbest < ftol:
        return 0
    if fbest <= ftol:
        return 1
    if B is not None and (np.abs(0.1 * sigma * B[:, axis] * D[axis]) < 1.0e-10).all():  # VD-CMA passes no B, D
        return -2
    if (0.2 * sigma * sd < 1.0e-10).any():
        return -3
    if D is not None and D.max() > 1.0e7 * D.min():
        return -4
    if it >= ilim:
        window = besthist[it - ilim : it + 1]
        if window.max() - window.min() < 1.0e-10:
            return -5
    if (sigma * sd > 1.0e3 * insigma).any():
        return -6
    if it > 2:
        joined = np.append(arfit, besthist)
        if joined.max() - joined.min() < 1.0e-12:
            return -7
    if (sigma * np.append(np.abs(pc), sd.max()) < 1.0e-11 * insigma).all():
        return -8
    return None


class _CmaRun:
    def __init__(self, fun_id, lower, upper, x0, maxiter, P, sigma, muperc, xtol, ftol, return_all, verbosity,
                 callback, rng, seed, eigh="host", workers=1, penalize=False):
        self.eigh = eigh
        self.penalize = penalize
        self.world = None
        if workers != 1:
            from ..parallel import require_world

            self.world = require_world(workers)
            self.world.shard(P)  # blocks of ceil(P / workers) rows, the last rank short
        self.fun_id, self.lower, self.upper, self.x0 = fun_id, lower, upper, x0
        self.maxiter, self.P, self.n = maxiter, P, len(lower)
        self.sigma0, self.muperc, self.xtol, self.ftol = sigma, muperc, xtol, ftol
        self.return_all, self.verbosity, self.callback = return_all, verbosity, callback
        self.rng, self.seed = rng, seed
        self.ctx = _device.Context()
        t = _device.torch()
        with t.cuda.stream(self.ctx.stream):
            self._run()

    def _run(self):
        ctx, L, n, P = self.ctx, self.ctx.L, self.n, self.P
        t = _device.torch()
        sp = ctx.stream_ptr
        ptr = _device.ptr
        stream = _rng.make_init_stream(self.rng, self.seed)
        key0, key1 = _rng.philox_key(self.seed) if self.rng == "philox" else (0, 0)

        # standardisation to [-1, 1]^n (cmaes/_cmaes.py:167-173)
        xm = 0.5 * (self.upper + self.lower)
        xstd = 0.5 * (self.upper - self.lower)

        def unstd(x):
            return x * xstd + xm

        d_xm, d_xstd = ctx.upload(xm), ctx.upload(xstd)
        xmean = stream.uniform(-1.0, 1.0, n) if self.x0 is None else (np.asarray(self.x0, dtype=np.float64) - xm) / xstd

        # strategy parameters (cmaes/_cmaes.py:184-205)
        mu = int(self.muperc * P)
        w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
        w /= w.sum()
        mueff = w.sum() ** 2 / np.square(w).sum()
        cc = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)
        cs = (mueff + 2.0) / (n + mueff + 5.0)
        c1 = 2.0 / ((n + 1.3) ** 2 + mueff)
        cmu = min(1.0 - c1, 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))
        damps = 1.0 + 2.0 * max(0.0, np.sqrt((mueff - 1.0) / (n + 1.0)) - 1.0) + cs
        chind = np.sqrt(n) * (1.0 - 1.0 / (4.0 * n) + 1.0 / (21.0 * n**2))

        pc = np.zeros(n)
        ps = np.zeros(n)
        B = np.eye(n)
        D = np.ones(n)
        invsqrtC = np.eye(n)
        diagC = np.ones(n)
        sigma = self.sigma0

        # device state
        d_w = ctx.upload(w)
        d_C = ctx.upload(np.eye(n))
        d_B = ctx.upload(B)
        d_D = ctx.upload(D)
        d_arx = ctx.empty((P, n))
        d_fit = ctx.empty((P,))
        # this rank's candidates: rows [row0, row0 + Pl) of the generation (all of them on one GPU)
        row0, Pl = (0, P) if self.world is None else self.world.shard(P)
        d_Z = ctx.empty((Pl, n))
        d_arx_loc = d_arx if self.world is None else ctx.empty((Pl, n))
        d_fit_loc = d_fit if self.world is None else ctx.empty((Pl,))
        d_xmean = ctx.upload(xmean)
        d_xold = ctx.empty((n,))
        d_pc = ctx.empty((n,))
        d_idx = ctx.empty((mu,), dtype=t.int64)
        d_Y = ctx.empty((mu, n))  # artmp scratch of the covariance update
        h_Z = t.empty((P, n), dtype=t.float64).pin_memory() if self.rng == "numpy-legacy" else None
        if self.penalize:
            bweights = _BoundaryWeights(n)
            d_v = ctx.empty((n,))
            d_pen = ctx.empty((P,))
            d_pen_loc = d_pen if self.world is None else ctx.empty((Pl,))
            xold = np.zeros(n)  # the reference reads an uninitialised array here in generation 1 (np.empty)

        def seen(rows):
            """What the caller sees of standardised candidates: with Penalize the clipped points
            (cmaes/_cmaes.py:238-256, 336-350), un-standardised."""
            return unstd(np.clip(rows, -1.0, 1.0)) if self.penalize else unstd(rows)

        if self.return_all:
            nout = int(np.ceil(self.verbosity * P))
            xall = np.empty((self.maxiter, max(1, nout), n))
            funall = np.empty((self.maxiter, max(1, nout)))

        nfev = 0
        eigeneval = 0
        eig, eig_sweeps, eig_warm = None, _CmaDeviceRun.COLD_SWEEPS, False  # launches beyond convergence are no-ops, but not free
        besthist = np.zeros(self.maxiter)
        ilim = int(10.0 + 30.0 * n / P)
        insigma = sigma
        it = 0
        while True:
            it += 1
            # ---- sample (device GEMM); normals from the numpy-legacy stream or in-kernel Philox ----
            if self.rng == "numpy-legacy":
                stream.randn(None, out=h_Z.numpy())  # P x randn(n), row by row == one block (cmaes/_cmaes.py:234)
                d_Z.copy_(h_Z[row0 : row0 + Pl], non_blocking=True)
            else:
                _lib.check(L.sx_cmaes_normals(ptr(d_Z), Pl, n, row0, it, key0, key1, sp), "sx_cmaes_normals")
            _lib.check(L.sx_cmaes_sample(ptr(d_xmean), sigma, ptr(d_B), ptr(d_D), ptr(d_Z), ptr(d_arx_loc), Pl, n, sp),
                       "sx_cmaes_sample")
            # ---- evaluate: fun(unstandardize(x)) fused (cmaes/_cmaes.py:173, 258) ----
            if not self.penalize:
                _common.evaluate_rows(ctx, self.fun_id, d_arx_loc, n, d_fit_loc, xm=d_xm, xstd=d_xstd)
            else:  # candidates clipped to the box before the objective (cmaes/_constraints.py:29-31)
                _common.evaluate_rows(ctx, self.fun_id, d_arx_loc, n, d_fit_loc, xm=d_xm, xstd=d_xstd, clip=True)
            if self.world is not None:  # every rank gets all candidates and fitness values back
                self.world.all_gather_rows(d_arx_loc, d_arx)
                self.world.all_gather_rows(d_fit_loc, d_fit)
            arfit = d_fit.cpu().numpy()
            if self.penalize:
                # host: boundary weights from the raw fitness spread (:33-76); device: weighted squared excess (:79)
                v = bweights.update(arfit, xmean, xold, sigma, diagC, mueff, it, P)
                if v.any():
                    d_v.copy_(t.from_numpy(np.ascontiguousarray(v)))
                    _common.penalty_rows(ctx, self.fun_id, d_arx_loc, n, d_xm, d_xstd, d_v, d_fit_loc, d_pen_loc)
                    if self.world is not None:
                        self.world.all_gather_rows(d_pen_loc, d_pen)
                    arfit = arfit + d_pen.cpu().numpy()
            nfev += P
            if self.return_all:
                if nout > 0:
                    xall[it - 1] = seen(d_arx[:nout].cpu().numpy())
                    funall[it - 1] = arfit[:nout]
                else:
                    k = int(arfit.argmin())
                    xall[it - 1] = seen(d_arx[k].cpu().numpy())
                    funall[it - 1] = arfit[k]
            # ---- rank and recombine (cmaes/_cmaes.py:272-277) ----
            order = np.argsort(arfit)
            d_idx.copy_(t.from_numpy(np.ascontiguousarray(order[:mu], dtype=np.int64)), non_blocking=False)
            xold = xmean
            d_xold.copy_(d_xmean)
            _lib.check(L.sx_cmaes_recombine(ptr(d_arx), ptr(d_idx), ptr(d_w), mu, n, ptr(d_xmean), sp),
                       "sx_cmaes_recombine")
            xmean = d_xmean.cpu().numpy()
            besthist[it - 1] = arfit[order[0]]
            # ---- evolution paths (cmaes/_cmaes.py:280-287), host vectors ----
            step = xmean - xold
            if self.eigh == "device":  # C^(-1/2) v = B ((B^T v) / D): no n^3 product on the host
                isc_step = np.dot(B, np.dot(B.T, step) / D)
            else:
                isc_step = np.dot(invsqrtC, step)
            ps = (1.0 - cs) * ps + np.sqrt(cs * (2.0 - cs) * mueff) * isc_step / sigma
            cond = np.linalg.norm(ps) / np.sqrt(1.0 - (1.0 - cs) ** (2.0 * nfev / P)) / chind < 1.4 + 2.0 / (n + 1.0)
            pc *= 1.0 - cc
            if cond:
                pc += np.sqrt(cc * (2.0 - cc) * mueff) * step / sigma
            # ---- covariance update on the device (cmaes/_cmaes.py:290-295) ----
            d_pc.copy_(t.from_numpy(pc), non_blocking=False)
            tmp_coef = 0.0 if cond else c1 * cc * (2.0 - cc)
            _lib.check(L.sx_cmaes_rank_mu(ptr(d_arx), ptr(d_idx), ptr(d_w), mu, ptr(d_xold), sigma, ptr(d_pc), c1, cmu,
                                          tmp_coef, ptr(d_C), ptr(d_Y), n, sp), "sx_cmaes_rank_mu")
            # ---- step size (cmaes/_cmaes.py:298) ----
            sigma *= np.exp((cs / damps) * (np.linalg.norm(ps) / chind - 1.0))
            # ---- eigendecomposition (cmaes/_cmaes.py:301-309): host LAPACK, as in the reference ----
            if nfev - eigeneval > P / (c1 + cmu) / n / 10.0:
                eigeneval = nfev
                _lib.check(L.sx_symmetrize_upper(ptr(d_C), n, sp), "sx_symmetrize_upper")
                if self.eigh == "device":
                    if eig is None:
                        eig = Eigh(ctx, n)
                    # ascending eigenvalues, eigenvectors in columns; from the second time on, started from the last ones
                    Dt, _ = eig(d_C, B=d_B, max_sweeps=eig_sweeps, start=d_B if eig_warm else None,
                                tol=max(1.0e-14, n * 1.1102230246251565e-16),  # LAPACK's own backward error, n * eps
                                refine=os.environ.get("SX_EIGH_REFINE", "1") != "0")  # as the device-resident loop does
                    was_warm, eig_warm = eig_warm, True
                    t.sqrt(Dt, out=d_D)
                    D = d_D.cpu().numpy()
                    used, ok, _off = eig.info()
                    if not was_warm:  # what the cold start needed says nothing about the warm ones
                        eig_sweeps = _CmaDeviceRun.WARM_SWEEPS0
                    else:
                        eig_sweeps = min(60, used + 1) if ok else 60
                    B = d_B.cpu().numpy()
                    diagC = d_C.diagonal().cpu().numpy()
                    status = _stop_status(it, n, self.maxiter, xmean, xold, besthist, arfit, order, sigma, insigma,
                                          ilim, pc, self.xtol, self.ftol, diagC, B, D)
                    if self.callback is not None:
                        res = OptimizeResult(x=seen(d_arx[int(order[0])].cpu().numpy()), fun=arfit[order[0]],
                                             nfev=nfev, nit=it)
                        if self.return_all:
                            res.update({"xall": xall[:it], "funall": funall[:it]})
                        self.callback(seen(d_arx.cpu().numpy()), res)
                    if status is not None:
                        break
                    continue
                Ch = d_C.cpu().numpy()
                D, B = self.eigh(Ch) if callable(self.eigh) else np.linalg.eigh(Ch)
                D, B = np.asarray(D, dtype=np.float64), np.asarray(B, dtype=np.float64)
                o = np.argsort(D)
                D = D[o]
                B = B[:, o]
                D = np.sqrt(D)
                invsqrtC = np.dot(np.dot(B, np.diag(1.0 / D)), B.T)
                d_B.copy_(t.from_numpy(np.ascontiguousarray(B)))
                d_D.copy_(t.from_numpy(np.ascontiguousarray(D)))
                diagC = np.diag(Ch).copy()
            else:
                diagC = d_C.diagonal().cpu().numpy()
            status = _stop_status(it, n, self.maxiter, xmean, xold, besthist, arfit, order, sigma, insigma, ilim, pc,
                                  self.xtol, self.ftol, diagC, B, D)
            if self.callback is not None:
                res = OptimizeResult(x=seen(d_arx[int(order[0])].cpu().numpy()), fun=arfit[order[0]], nfev=nfev, nit=it)
                if self.return_all:
                    res.update({"xall": xall[:it], "funall": funall[:it]})
                self.callback(seen(d_arx.cpu().numpy()), res)
            if status is not None:
                break

        res = OptimizeResult(
            x=seen(d_arx[int(order[0])].cpu().numpy()),
            success=status >= 0,
            status=status,
            message=_common.messages[status],
            fun=arfit[order[0]],
            nfev=nfev,
            nit=it,
        )
        if self.return_all:
            res.update({"xall": xall[:it], "funall": funall[:it]})
        if self.rng == "numpy-legacy":
            stream.sync_back()
        ctx.sync()
        self._res = res

    def result(self):
        return self._res


register("cmaes", minimize)
