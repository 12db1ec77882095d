"""Neighbourhood Algorithm front end + generation loop for ``backend="hip"``.

Reference: stochopy/optimize/na/_na.py:11-128 (``minimize``), :131-262 (``na`` loop), :265-305 (``mutation``).
Every generation resamples ``popsize`` points inside the Voronoi cells of the ``nr`` best of ALL models sampled so far;
the walk costs O(popsize * models * ndim) and runs on the device (csrc/sx_na.hip, models stored column-major), as do the
objective, the greedy selection and the best/termination step (the kernels the other optimizers use).  On the host:
the ranking of the models (``np.argsort`` -- the reference's own call, so its tie order), and the scalar recurrence
``d1`` of the walk (na/_na.py:298-300).  That recurrence is numpy SCALAR arithmetic in the reference, where ``** 2`` is
libm ``pow`` -- not always the correctly rounded square -- and the walk is chaotic in those bits; evaluating it with the
same libm between two axis steps (``popsize`` numbers down, ``popsize`` numbers up per axis and generation) is what
makes a run reproduce the reference bit for bit (numpy-legacy draws; ``+ - *`` objectives).
"""
import numpy as np

from .. import _device, _lib, _rng
from . import _common
from ._helpers import OptimizeResult, register

__all__ = ["minimize"]

_MAX_BYTES = 64 << 30  # model store + cell distances kept resident (288 GB HBM; a guard against runaway sizes)


def minimize(
    fun,
    bounds,
    x0=None,
    args=(),
    maxiter=100,
    popsize=10,
    nrperc=0.5,
    seed=None,
    xtol=1.0e-8,
    ftol=1.0e-8,
    workers=1,
    backend=None,
    return_all=False,
    verbosity=1.0,
    callback=None,
    rng=None,
    host_workers=None,
    host_backend=None,
):
    """Minimize an objective function using the Neighbourhood Algorithm on MI355X (reference na/_na.py:11-27; its
    ``callback=True`` default -- which makes a direct call raise -- is not copied: ``None`` here, as through
    ``stochopy.optimize.minimize``).  One GPU (``workers=1``): the model store is shared by all walks."""
    fun_id = _common.resolve_objective(fun, args, workers, backend, host_workers, host_backend)
    lower, upper = _common.as_bounds(bounds)
    if x0 is not None:
        if np.ndim(x0) != 2 or np.shape(x0)[1] != len(bounds):
            raise ValueError()
    if popsize < 2:
        raise ValueError()
    if x0 is not None and len(x0) != popsize:
        raise ValueError()
    if not 0.0 < nrperc <= 1.0:
        raise ValueError()
    if callback is not None and not hasattr(callback, "__call__"):
        raise ValueError()
    _common.resolve_backend(backend, fun_id)
    rng = _common.resolve_rng(rng)
    if popsize > 65535:
        raise ValueError("method 'na': popsize <= 65535 (one workgroup row of the cell-walk kernels per sample)")
    if _common.resolve_workers(workers, fun_id) != 1:  # (the reference's NA takes workers like every method: na/_na.py:131)
        _common.replicated_workers("na", workers, "every walk reads the whole model store, which lives on one GPU")
    return _NaRun(fun_id, lower, upper, x0, int(maxiter), int(popsize), float(nrperc), float(xtol), float(ftol),
                  bool(return_all), float(verbosity), callback, rng, seed).result()


class _NaRun:
    def __init__(self, fun, lower, upper, x0, maxiter, P, nrperc, xtol, ftol, return_all, verbosity, callback, rng, seed):
        n = len(lower)
        cap = max(2, maxiter) * P  # (generation 2 always runs before the maxiter test, as in the reference: na/_na.py:196-263)
        if 8 * cap * (n + P) > _MAX_BYTES:
            raise ValueError(f"method 'na': maxiter * popsize = {cap} models need {8 * cap * (n + P) / 2**30:.0f} GiB "
                             "for the model store and the cell distances (limit 64 GiB)")
        self.ctx = ctx = _device.Context()
        t = _device.torch()
        with t.cuda.stream(ctx.stream):
            self._run(fun, lower, upper, x0, maxiter, P, n, cap, nrperc, xtol, ftol, return_all, verbosity, callback,
                      rng, seed)

    def _run(self, fun, lower, upper, x0, maxiter, P, n, cap, nrperc, xtol, ftol, return_all, verbosity, callback, rng, seed):
        ctx, L, ptr, t = self.ctx, self.ctx.L, _device.ptr, _device.torch()
        sp = ctx.stream_ptr
        stream = _rng.make_init_stream(rng, seed)
        key0, key1 = _rng.philox_key(seed) if rng == "philox" else (0, 0)
        # normalisation to the unit cube (na/_na.py:149-157); fixed axes (upper == lower) carry `upper`
        span = upper - lower
        free = span > 0.0
        span = np.where(free, span, 1.0)
        normalize = lambda x: np.where(free, (x - lower) / span, upper)  # noqa: E731
        unnormalize = lambda x: np.where(free, x * span + lower, upper)  # noqa: E731
        d_lower, d_span = ctx.upload(lower), ctx.upload(span)
        d_fixed = ctx.upload((~free).astype(np.int32))
        nr = max(1, int(nrperc * P))                                                # :160
        if x0 is not None:
            X0 = np.asarray(x0, dtype=np.float64)
        elif rng == "philox":  # the counter-based Latin hypercube of the in-kernel mode (drawn on the device, _rng.py)
            X0 = _rng.philox_latin_hypercube(ctx, ctx.empty((P, n)), 0, P, ctx.upload(lower), ctx.upload(upper),
                                             seed).cpu().numpy()
        else:
            X0 = stream.latin_hypercube(P, n, lower, upper)
        Xn = np.ascontiguousarray(normalize(X0))
        # device state: the samples being built, personal bests, the model store (column-major) and cell distances
        d_X = ctx.upload(Xn)
        d_pbest = d_X.clone()
        d_XT = ctx.empty((n, cap))
        d_XT[:, :P].copy_(d_X.t())
        d_d2 = ctx.empty((P, cap))
        d_cand = ctx.empty((P,))
        d_pbestfit = ctx.empty((P,))
        d_kidx = ctx.empty((P,), dtype=t.int64)
        d_u, d_d1, d_xnew = ctx.empty((P, n)), ctx.empty((P,)), ctx.empty((P,))
        d_ws = ctx.empty((2 * P * 64,))
        npart = int(L.sx_num_partials(P, n))
        part_f, part_i = ctx.empty((npart,)), ctx.empty((npart,), dtype=t.int64)
        h_u = t.empty((P, n), dtype=t.float64).pin_memory() if rng == "numpy-legacy" else None

        # fun(unnormalize(X)) (:165-169).  A fixed axis carries `upper` in the normalised initial population and is put
        # back to `upper` by unnormalize's np.where; the device un-normalisation x * span + lower wants a 0 there
        d_eval = d_X.clone()
        if not free.all():
            d_eval[:, t.from_numpy(~free).to(ctx.device)] = 0.0
        _common.evaluate_rows(ctx, fun, d_eval, n, d_pbestfit, xm=d_lower, xstd=d_span)
        pfit = d_pbestfit.cpu().numpy()
        g = int(np.argmin(pfit))
        d_gbest = d_X[g].clone()
        st = _lib.SxState(it=1, gbidx=g, gfit=float(pfit[g]), dx=0.0, status=_lib.SX_STATUS_NONE, done=0)
        d_state = ctx.upload(np.frombuffer(bytes(st), dtype=np.int64).copy())
        models = np.empty((cap, n))          # host copy of the store, store order (generation by generation)
        models[:P] = Xn
        fit_ref = pfit.copy()                # fitness of all models in the REFERENCE's order (newest generation first)
        M = P
        if return_all:
            nout = int(np.ceil(verbosity * P))
            xall = np.empty((max(2, maxiter), max(1, nout), n))
            funall = np.empty((max(2, maxiter), max(1, nout)))
            if nout > 0:  # NB the reference stores the NORMALISED rows at iteration 1 (:185-193)
                xall[0], funall[0] = Xn[:nout], pfit[:nout]
            else:
                xall[0], funall[0] = Xn[g], pfit[g]
        gbest, gfit = Xn[g].copy(), pfit[g]

        def report(it, Xcur):
            res = OptimizeResult(x=unnormalize(gbest), fun=gfit, nfev=it * P, nit=it)
            if return_all:
                res.update({"xall": xall[:it], "funall": funall[:it]})
            callback(unnormalize(Xcur), res)

        if callback is not None:
            report(1, Xn)
        free_axes = [j for j in range(n) if free[j]]
        it, status = 1, None
        while status is None:
            it += 1
            # ---- cells: the nr best of all models, the reference's ranking call (:274) ----
            order = fit_ref.argsort()[:nr]
            ref_idx = order[np.arange(P) % nr]                    # k = ix[i % nr]
            G = M // P                                            # generations stored so far
            slots = (G - 1 - ref_idx // P) * P + ref_idx % P      # reference index (newest first) -> store position
            d_kidx.copy_(t.from_numpy(np.ascontiguousarray(slots, dtype=np.int64)))
            centre = models[slots]                                # (P, n) host copy of the cell centres
            # ---- the uniforms behind np.random.uniform(low, high) (:296), sample-major like the reference ----
            if rng == "numpy-legacy":
                hu = h_u.numpy()
                hu[:, free] = stream.random((P, len(free_axes)))
                d_u.copy_(h_u, non_blocking=True)
            else:
                _lib.check(L.sx_na_uniforms(ptr(d_u), P, n, it, key0, key1, sp), "sx_na_uniforms")
            # ---- the walks (:275-303): one device step per free axis, the scalar d1 recurrence in between ----
            _lib.check(L.sx_na_begin(ptr(d_XT), cap, M, n, ptr(d_kidx), P, ptr(d_X), ptr(d_d2), sp), "sx_na_begin")
            d1 = [0.0] * P
            jp = -1
            for j in free_axes:
                d_d1.copy_(t.from_numpy(np.array(d1, dtype=np.float64)))
                _lib.check(L.sx_na_axis(ptr(d_XT), cap, M, n, j, jp, ptr(d_kidx), P, ptr(d_u), ptr(d_d1), ptr(d_X),
                                        ptr(d_d2), ptr(d_ws), ptr(d_xnew), sp), "sx_na_axis")
                jp = j
                if j < n - 1:
                    xnew = d_xnew.cpu().numpy()
                    cj = centre[:, j]
                    zero = np.float64(0.0)  # (Xall[k, j+1] - X[i, j+1]): the next axis still sits on the centre
                    # numpy SCALAR arithmetic, as in the reference (:298-300): `** 2` on a scalar is libm pow
                    d1 = [d1[i] + ((cj[i] - xnew[i]) ** 2 - zero ** 2) for i in range(P)]
            _lib.check(L.sx_na_commit(ptr(d_X), P, n, ptr(d_fixed), ptr(d_XT), cap, M, sp), "sx_na_commit")
            # ---- objective + greedy selection + best / termination (_common.py:123-160) ----
            _common.evaluate_rows(ctx, fun, d_X, n, d_cand, xm=d_lower, xstd=d_span)
            _lib.check(L.sx_rows_select(ptr(d_X), n, ptr(d_cand), ptr(d_pbest), ptr(d_pbest), n, ptr(d_pbestfit), None, P, n,
                                        ptr(d_state), ptr(part_f), ptr(part_i), sp), "sx_rows_select")
            _lib.check(L.sx_select_finalize(ptr(part_f), ptr(part_i), npart, ptr(d_pbest), ptr(d_pbest), n, n, ptr(d_gbest),
                                            ptr(d_state), maxiter, xtol, ftol, sp), "sx_select_finalize")
            Xcur = d_X.cpu().numpy()
            pfit = d_cand.cpu().numpy()
            state = ctx.read_state(d_state)
            models[M : M + P] = Xcur
            fit_ref = np.concatenate((pfit, fit_ref))                               # :223-224
            M += P
            gbest, gfit = d_gbest.cpu().numpy(), state.gfit
            if return_all:
                if nout > 0:
                    xall[it - 1], funall[it - 1] = unnormalize(Xcur[:nout]), pfit[:nout]
                else:
                    k = int(pfit.argmin())
                    xall[it - 1], funall[it - 1] = unnormalize(Xcur[k]), pfit[k]
            if state.status != _lib.SX_STATUS_NONE:
                status = int(state.status)
            if callback is not None:
                report(it, Xcur)
        res = OptimizeResult(x=unnormalize(gbest), success=status >= 0, status=status, message=_common.messages[status],
                             fun=gfit, nfev=it * P, nit=it)
        if return_all:
            res.update({"xall": xall[:it], "funall": funall[:it]})
        if rng == "numpy-legacy":
            stream.sync_back()
        ctx.sync()
        self._res = res

    def result(self):
        return self._res


register("na", minimize)
