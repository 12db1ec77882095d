"""VD-CMA front end + generation loop for ``backend="hip"``.

Reference: stochopy/optimize/vdcma/_vdcma.py:12-141 (``minimize``) and :144-425 (``vdcma`` loop), :428-460
(the moment and natural-gradient helpers); stopping rules shared with CMA-ES (cmaes/_cmaes.py:360-434 without
B and D).  SURVEY.md section 8f rank 3: the covariance model is D (I + v v^T) D, so a generation is O(P n):

* on the device (csrc/sx_cmaes.hip ``vd_sample_kernel``, csrc/sx_core.hip): the P candidates
  ``y = d o (z + (sqrt(1+|v|^2)-1)(z.vn) vn)``, ``x = xmean + sigma y`` (one wavefront each, normals from the
  numpy-legacy stream or in-kernel Philox), the objective with fused un-standardisation and -- with
  ``constraints="Penalize"`` -- the clipping and the weighted squared excess (shared with CMA-ES);
* on the host, as in the reference: ranking, the weighted sums over the mu selected rows (moments p, q of
  :428-444), the natural-gradient step for v and d (:447-460), step size, stopping rules -- O(mu n) numpy.

``workers > 1``: candidates sharded by rows like CMA-ES (one all-gather of y, x and the fitness per generation).  The sharded
run is the one-GPU run bit for bit for models of up to 2048 coordinates.  Longer models (csrc/sx_wide.hip): on one GPU the
candidates kernel leaves t_k = sum(yd * vn) as a by-product, with yd the step BEFORE it is rounded through y = d * yd; the
sharded run forms t_k from the gathered y as (y / d) . vn (csrc/sx_vd_loop.hip vd_t_kernel, what the oracle does): the two
differ in the last bits of every t_k, so there the sharded run follows the one-GPU run to rounding, not bit for bit.
"""

import os

import numpy as np

from .. import _device, _lib, _rng
from . import _common
from ._cmaes import _BoundaryWeights, _stop_status
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
    host_workers=None,
    host_backend=None,
):
    """Minimize an objective function using VD-CMA on MI355X (reference vdcma/_vdcma.py:12-30)."""
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
    if (rng == "philox" and isinstance(fun_id, int) and os.environ.get("SX_CMA_LOOP", "") != "host"
            and (constraints is None or 20.0 + 3.0 * len(lower) / int(popsize) + 1.0 <= 256.0)):
        # nothing the host has to see between generations: the whole loop (and the history) stays on the device --
        # since round 3 including constraints="Penalize" (boundary-weight bookkeeping shared with CMA-ES: cma_penalty_kernel)
        # and workers > 1 (own candidates, one gather of steps / candidates / fitness, the O(n) model update replicated);
        # a callback is served from this loop too (the host then looks at every generation).  SX_CMA_LOOP=host: the
        # host-driven loop (tests)
        return _VdDeviceRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(sigma), float(muperc),
                            float(xtol), float(ftol), seed, bool(return_all), float(verbosity),
                            penalize=constraints == "Penalize", workers=workers, callback=callback).result()
    run = _VdRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(sigma), float(muperc), float(xtol),
                 float(ftol), bool(return_all), float(verbosity), callback, rng, seed, workers,
                 constraints == "Penalize")
    return run.result()


def _strategy_constants(n, P, muperc):
    """Selection weights and learning rates (vdcma/_vdcma.py:185-199)."""
    mu = int(muperc * P)
    w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    w /= w.sum()
    mueff = w.sum() ** 2 / np.square(w).sum()
    cc = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)
    cfactor = (n - 5.0) / 6.0
    c1 = cfactor * 2.0 / ((n + 1.3) ** 2 + mueff)
    cmu = min(1.0 - c1, cfactor * 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))
    return mu, w, mueff, cc, c1, cmu


class _VdDeviceRun:
    """The reference's loop (vdcma/_vdcma.py:232-425) with every per-generation step on the device
    (csrc/sx_cma_loop.hip, sx_vdcma_generation): the host enqueues generations and looks at the 128-byte state every
    LOOK of them.  Draws: Philox normals keyed by (seed, generation, row); the injection's normal row is "row P"."""

    LOOK = 16

    def __init__(self, fun_id, lower, upper, x0, maxiter, P, sigma, muperc, xtol, ftol, seed, return_all=False,
                 verbosity=1.0, run=True, penalize=False, workers=1, callback=None):
        """``run=False`` only builds the device state (``self.buffers``, ``self.args``): tests drive single generations
        with ``step`` from a state of their choosing."""
        import ctypes as C

        ctx = self.ctx = _device.Context()
        t = _device.torch()
        ptr, n = _device.ptr, len(lower)
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
            mu, w, mueff, cc, c1, cmu = _strategy_constants(n, P, muperc)
            vvec = init.randn(n) / np.sqrt(n)  # the first direction comes right after the initial mean in the stream
            with _common.host_blas_single_thread():  # (a threaded BLAS wakes 64+ spinning threads for this one dot product)
                norm_v2 = float(np.dot(vvec, vvec))
            norm_v = float(np.sqrt(norm_v2))
            # wide models whose candidates nobody else wants (no callback, history, Penalize, sharding): x = xmean + sigma y
            # is not kept -- the moments kernel forms it again from y, bit for bit (sx_vd_args.arx NULL): a quarter of a
            # generation's memory traffic
            keep_x = not (run and n > _lib.NARROW_DIM and callback is None and not return_all and not penalize and world is None
                          and os.environ.get("SX_VD_KEEP_X", "0") != "1")
            keep = self.buffers = dict(
                Z=ctx.empty((P, n)), ary=ctx.empty((P, n)), fit=ctx.empty((P,)),
                xmean=ctx.upload(xmean), xold=ctx.zeros((n,)), dx=ctx.zeros((n,)), dvec=ctx.upload(np.ones(n)),
                vvec=ctx.upload(vvec), vn=ctx.upload(vvec / norm_v), pc=ctx.zeros((n,)), zinj=ctx.empty((n,)),
                dy=ctx.zeros((n,)), w=ctx.upload(w), mws=ctx.empty((((mu + 7) // 8) * 8 + 4 * 64 * n,)),
                mout=ctx.empty((4, n)), besthist=ctx.zeros((maxiter,)), xm=ctx.upload(xm), xstd=ctx.upload(xstd),
                xbest=ctx.zeros((n,)), order=ctx.empty((P,), dtype=t.int64))
            if keep_x:
                keep["arx"] = ctx.empty((P, n))
            if penalize:  # cmaes/_constraints.py:4-82 on the device: weights 0, spread history [1.0], both phase flags as at the start
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
            st.reserved[0], st.reserved[1], st.reserved[2] = 0.0, norm_v2, norm_v           # ps, |v|^2, |v|
            st.reserved[3], st.reserved[4] = 0.0, float(np.sqrt(1.0 + norm_v2) - 1.0)      # injection off, coefficient
            d_state = keep["state"] = ctx.upload(np.frombuffer(bytes(st), dtype=np.float64))
            a = _lib.SxVdArgs(**{k: ptr(v) for k, v in keep.items()})
            a.P, a.hist_rows, a.n, a.mu, a.fun_id, a.maxiter = P, nout, n, mu, fun_id, maxiter
            a.ilim = int(10 + 30 * n / P)
            a.cs, a.ds, a.cc, a.c1, a.cmu, a.mueff, a.wsum = 0.3, float(np.sqrt(n)), cc, c1, cmu, mueff, float(w.sum())
            a.xtol, a.ftol, a.insigma, a.key0, a.key1 = xtol, ftol, sigma, key0, key1
            self.args, self.P = a, P
            if not run:
                return
            look, since = 1, 0
            cb_pin = cb_hist = None
            state = st
            if world is not None:
                ary_loc, arx_loc, fit_loc = ctx.empty((Pl, n)), ctx.empty((Pl, n)), ctx.empty((Pl,))
            for gen in range(1, maxiter + 1):
                if world is None:
                    _lib.check(ctx.L.sx_vdcma_generation(C.byref(a), gen, ctx.stream_ptr), "sx_vdcma_generation")
                else:  # own candidates, one gather of steps / candidates / fitness, the model update replicated
                    _lib.check(ctx.L.sx_vdcma_generation_stage(C.byref(a), gen, 0, row0, Pl, ptr(ary_loc), ptr(arx_loc),
                                                               ptr(fit_loc), ctx.stream_ptr), "sx_vdcma_generation_stage")
                    world.all_gather_rows(ary_loc, keep["ary"])
                    world.all_gather_rows(arx_loc, keep["arx"])
                    world.all_gather_rows(fit_loc, keep["fit"])
                    _lib.check(ctx.L.sx_vdcma_generation_stage(C.byref(a), gen, 1, 0, 0, None, None, None, ctx.stream_ptr),
                               "sx_vdcma_generation_stage")
                since += 1
                if since >= look or gen == maxiter:
                    state = _lib.SxCmaState.from_buffer_copy(d_state.cpu().numpy().tobytes())
                    if callback is not None:
                        # what the reference hands over (vdcma/_vdcma.py:413-423): all candidates (the clipped ones with
                        # Penalize), un-standardised, and the best of them; with return_all the history so far
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
                    # looks get rarer (1, 2, 4, ... LOOK generations apart) whatever a generation costs: after a stop the
                    # generations already enqueued are no-ops, and the interval no longer depends on the host's clock
                    # (sharded: every generation, so that all ranks stop enqueueing collectives together)
                    if world is None and callback is None and look < self.LOOK:
                        look *= 2
                    since = 0
            if not state.done:  # cannot happen: generation maxiter sets status -1
                raise RuntimeError("VD-CMA device loop ended without a status")
            if state.status == -99:  # csrc/sx_vd_loop.hip kVwFault
                raise RuntimeError("VD-CMA model update: a grid-wide wait of the single-launch chain ran out (is another kernel "
                                   "holding the device?); SX_VD_CHAIN=0 runs the chain as one launch per phase")
            nit = int(state.stop_it)
            self._res = OptimizeResult(x=keep["xbest"].cpu().numpy(), success=state.status >= 0, status=int(state.status),
                                       message=_common.messages[int(state.status)], fun=float(state.fbest),
                                       nfev=nit * P, nit=nit)
            if return_all:
                self._res.update({"xall": keep["hist_x"][:nit].cpu().numpy(), "funall": keep["hist_f"][:nit].cpu().numpy()})
            ctx.sync()

    def step(self, gen):
        import ctypes as C

        t = _device.torch()
        with t.cuda.stream(self.ctx.stream):
            _lib.check(self.ctx.L.sx_vdcma_generation(C.byref(self.args), int(gen), self.ctx.stream_ptr),
                       "sx_vdcma_generation")

    def read_state(self):
        return _lib.SxCmaState.from_buffer_copy(self.buffers["state"].cpu().numpy().tobytes())

    def result(self):
        return self._res


def _moments(vn, norm_v2, y, w=None):
    """Moments p, q of the selected steps under the current model (vdcma/_vdcma.py:428-444)."""
    t = np.dot(y, vn)
    shrink = norm_v2 / (1.0 + norm_v2)
    if w is None:
        return y**2 - shrink * (t * y * vn) - 1.0, t * y - (0.5 * (t**2 + 1.0 + norm_v2)) * vn
    p = np.dot(w, y**2 - shrink * (t[:, None] * (y * vn)) - 1.0)
    q = np.dot(w, t[:, None] * y - np.outer(0.5 * (t**2 + 1.0 + norm_v2), vn))
    return p, q


def _natural_gradient(dvec, vn, vnn, norm_v, norm_v2, alpha, avec, bsca, invavnn, p, q):
    """Steps for v and d (vdcma/_vdcma.py:447-460)."""
    r = p - alpha / (1.0 + norm_v2) * ((2.0 + norm_v2) * q * vn - norm_v2 * np.dot(vn, q) * vnn)
    s = r / avec - bsca * np.dot(r, invavnn) / (1.0 + bsca * np.dot(vnn, invavnn)) * invavnn
    ngv = q / norm_v - alpha / norm_v * ((2.0 + norm_v2) * (vn * s) - np.dot(s, vnn) * vn)
    return ngv, dvec * s


class _VdRun:
    def __init__(self, fun_id, lower, upper, x0, maxiter, P, sigma, muperc, xtol, ftol, return_all, verbosity,
                 callback, rng, seed, workers, penalize):
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
        xm = 0.5 * (self.upper + self.lower)
        xstd = 0.5 * (self.upper - self.lower)

        def seen(rows):  # what the caller sees: un-standardised, clipped to the box with Penalize
            return (np.clip(rows, -1.0, 1.0) if self.penalize else rows) * xstd + xm

        d_xm, d_xstd = ctx.upload(xm), ctx.upload(xstd)
        xmean = stream.uniform(-1.0, 1.0, n) if self.x0 is None else (np.asarray(self.x0, dtype=np.float64) - xm) / xstd
        xold = np.zeros(n)  # the reference leaves this uninitialised until the first update

        # selection weights and learning rates (vdcma/_vdcma.py:185-199)
        mu = int(self.muperc * P)
        w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
        w /= w.sum()
        mueff = w.sum() ** 2 / np.square(w).sum()
        cc = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)
        cfactor = (n - 5.0) / 6.0
        c1 = cfactor * 2.0 / ((n + 1.3) ** 2 + mueff)
        cmu = min(1.0 - c1, cfactor * 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))

        # dynamic state (:202-213); the first direction comes right after the initial mean in the stream
        inject = False
        cs, ds = 0.3, np.sqrt(n)
        dx = np.zeros(n)
        ps = 0.0
        dvec = np.ones(n)
        vvec = stream.randn(n) / np.sqrt(n)
        norm_v2 = np.dot(vvec, vvec)
        norm_v = np.sqrt(norm_v2)
        vn = vvec / norm_v
        vnn = vn**2
        pc = np.zeros(n)
        sigma = self.sigma0

        row0, Pl = (0, P) if self.world is None else self.world.shard(P)
        d_Z = ctx.empty((Pl, n))
        d_ary = ctx.empty((P, n))
        d_arx = ctx.empty((P, n))
        d_fit = ctx.empty((P,))
        d_ary_loc = d_ary if self.world is None else ctx.empty((Pl, n))
        d_arx_loc = d_arx if self.world is None else ctx.empty((Pl, n))
        d_fit_loc = d_fit if self.world is None else ctx.empty((Pl,))
        d_dvec, d_vn, d_xmean, d_dy = ctx.empty((n,)), ctx.empty((n,)), ctx.empty((n,)), ctx.empty((n,))
        d_zinj = ctx.empty((1, n))
        d_sel = ctx.empty((mu,), dtype=t.int64)
        d_w = ctx.upload(w)
        d_mws = ctx.empty((((mu + 7) // 8) * 8 + 4 * 64 * n,))
        d_mout = ctx.empty((4, n))
        h_Z = t.empty((P, n), dtype=t.float64).pin_memory() if self.rng == "numpy-legacy" else None
        if self.penalize:
            bweights = _BoundaryWeights(n)
            d_v = ctx.empty((n,))
            d_pen = ctx.empty((P,))
            d_pen_loc = d_pen if self.world is None else ctx.empty((Pl,))
        if self.return_all:
            nout = int(np.ceil(self.verbosity * P))
            xall = np.empty((self.maxiter, max(1, nout), n))
            funall = np.empty((self.maxiter, max(1, nout)))

        def up(dst, a):
            dst.copy_(t.from_numpy(np.ascontiguousarray(a)))

        nfev = 0
        besthist = np.zeros(self.maxiter)
        ilim = int(10 + 30 * n / P)
        insigma = sigma
        it = 0
        while True:
            it += 1
            # ---- candidates (vdcma/_vdcma.py:236-248): normals, then the O(n) model on the device ----
            if self.rng == "numpy-legacy":
                stream.randn(None, out=h_Z.numpy())
                d_Z.copy_(h_Z[row0 : row0 + Pl], non_blocking=True)
            else:
                _lib.check(L.sx_cmaes_normals(ptr(d_Z), Pl, n, row0, it, key0, key1, sp), "sx_cmaes_normals")
            dy = None
            if inject:  # mean-shift injection: rows 0 and 1 become +-dy (:241-247)
                ddx = dx / dvec
                mnorm = (ddx**2).sum() - np.dot(ddx, vvec) ** 2 / (1.0 + norm_v2)
                if self.rng == "numpy-legacy":
                    zinj = stream.randn(n)
                else:  # "row P" of the generation's normals, one past the population
                    _lib.check(L.sx_cmaes_normals(ptr(d_zinj), 1, n, P, it, key0, key1, sp), "sx_cmaes_normals")
                    zinj = d_zinj.cpu().numpy()[0]
                dy = np.linalg.norm(zinj) / np.sqrt(mnorm) * dx
                up(d_dy, dy)
            up(d_dvec, dvec)
            up(d_vn, vn)
            up(d_xmean, xmean)
            _lib.check(L.sx_vdcma_sample(ptr(d_Z), Pl, n, row0, ptr(d_dvec), ptr(d_vn), float(np.sqrt(1.0 + norm_v2) - 1.0),
                                         ptr(d_xmean), float(sigma), ptr(d_dy) if dy is not None else None,
                                         ptr(d_ary_loc), ptr(d_arx_loc), sp), "sx_vdcma_sample")
            diagC = (dvec * (1.0 + vvec * vvec)) * dvec  # diag of D (I + v v^T) D (:249-254)
            # ---- objective (+ Penalize), as in CMA-ES ----
            if not self.penalize:
                _common.evaluate_rows(ctx, self.fun_id, d_arx_loc, n, d_fit_loc, xm=d_xm, xstd=d_xstd)
            else:
                _common.evaluate_rows(ctx, self.fun_id, d_arx_loc, n, d_fit_loc, xm=d_xm, xstd=d_xstd, clip=True)
            if self.world is not None:
                self.world.all_gather_rows(d_ary_loc, d_ary)
                self.world.all_gather_rows(d_arx_loc, d_arx)
                self.world.all_gather_rows(d_fit_loc, d_fit)
            arfit = d_fit.cpu().numpy()
            if self.penalize:
                v = bweights.update(arfit, xmean, xold, sigma, diagC, mueff, it, P)
                if v.any():
                    up(d_v, v)
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
            # ---- rank, mean shift (:289-295): the mu selected rows of y and x come to the host ----
            # ---- rank (host argsort), then every O(mu n) sum of the update on the device: the weighted means of the
            # selected x and y and the rank-mu moments (:289-295, :317, :331-339); four n-vectors come back ----
            order = np.argsort(arfit)
            d_sel.copy_(t.from_numpy(np.ascontiguousarray(order[:mu], dtype=np.int64)))
            _lib.check(L.sx_vdcma_moments(ptr(d_arx), ptr(d_ary), ptr(d_sel), ptr(d_w), mu, n, ptr(d_dvec), ptr(d_vn),
                                          float(norm_v2), ptr(d_mws), ptr(d_mout), sp), "sx_vdcma_moments")
            wx, wy, p_mu_dev, q_mu_dev = d_mout.cpu().numpy()
            dx = wx - w.sum() * xmean
            xold = xmean.copy()
            xmean = xmean + dx
            besthist[it - 1] = arfit[order[0]]
            # ---- step size from the rank gap of the injected pair (:298-306) ----
            if inject:
                gap = (np.where(order == 1)[0][0] - np.where(order == 0)[0][0]) / (P - 1.0)
                ps += cs * (gap - ps)
                sigma *= np.exp(ps / ds)
                cond = ps < 0.5
            else:
                inject = True
                cond = True
            # ---- evolution path, model constants (:309-328) ----
            pc *= 1.0 - cc
            if cond:
                pc += np.sqrt(cc * (2.0 - cc) * mueff) * wy
            gamma = 1.0 / np.sqrt(1.0 + norm_v2)
            alpha = np.sqrt(norm_v2**2 + (1.0 + norm_v2) / vnn.max() * (2.0 - gamma)) / (2.0 + norm_v2)
            if alpha < 1.0:
                beta = (4.0 - (2.0 - gamma) / vnn.max()) / (1.0 + 2.0 / norm_v2) ** 2
            else:
                alpha, beta = 1.0, 0.0
            bsca = 2.0 * alpha**2 - beta
            avec = 2.0 - (bsca + 2.0 * alpha**2) * vnn
            invavnn = vnn / avec
            # ---- moments, natural gradient, update of v and d (:331-378) ----
            p_mu, q_mu = (np.zeros(n), np.zeros(n)) if cmu == 0.0 else (p_mu_dev, q_mu_dev)
            p_one, q_one = (np.zeros(n), np.zeros(n)) if c1 == 0.0 else _moments(vn, norm_v2, pc / dvec)
            p = cmu * p_mu
            q = cmu * q_mu
            if cond:
                p = p + c1 * p_one
                q = q + c1 * q_one
            if cmu + c1 > 0.0:
                ngv, ngd = _natural_gradient(dvec, vn, vnn, norm_v, norm_v2, alpha, avec, bsca, invavnn, p, q)
                step = min(1.0, 0.7 * norm_v / np.sqrt(np.dot(ngv, ngv)), 0.7 * (dvec / np.abs(ngd)).min())
                vvec = vvec + step * ngv
                dvec = dvec + step * ngd
            norm_v2 = np.dot(vvec, vvec)
            norm_v = np.sqrt(norm_v2)
            vn = vvec / norm_v
            vnn = vn**2
            status = _stop_status(it, n, self.maxiter, xmean, xold, besthist, arfit, order, sigma, insigma, ilim, pc,
                                  self.xtol, self.ftol, diagC, None, None)
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


register("vdcma", minimize)
