"""Shared pieces of the per-method front ends.

Reference: stochopy/optimize/_common.py:13-24 (messages), :27-106 (the backend
hook this package sits beside).  Where the reference's hook wraps the scalar
objective into a population-level callable and parallelises only those calls,
``backend="hip"`` moves the whole generation onto the GPU; the hook contract that
is kept is the user-visible one: same ``minimize`` options, synchronous updating
whenever a parallel backend is chosen (de/_de.py:142-145), identical results for
a given seed (rng="numpy-legacy").
"""
import warnings

import numpy as np

from ..factory.benchmark import Objective

messages = {
    -8: "TolX",
    -7: "TolFun",
    -6: "TolXUp",
    -5: "EqualFunValues",
    -4: "ConditionCov",
    -3: "NoEffectCoord",
    -2: "NoEffectAxis",
    -1: "maximum number of iterations is reached",
    0: "best solution changes less than xtol",
    1: "best solution value is lower than ftol",
}

RNG_MODES = ("numpy-legacy", "philox")


def resolve_backend(backend, fun_id=None):
    """`backend="hip"` (or None: this package has only that one).  With a plain Python objective the reference's
    "threading" / "loky" name the HOST pool that evaluates it (resolve_objective); the generation still runs on the GPU."""
    if backend in (None, "hip"):
        return "hip"
    if getattr(fun_id, "from_reference_options", False):
        return "hip"
    if backend in ("loky", "threading", "mpi"):
        raise ValueError(f"backend '{backend}' belongs to keurfonluu/stochopy (CPU); stochopy_amd only provides 'hip'")
    raise ValueError(f"unknown backend '{backend}'")  # reference _common.py:74-75


class External:
    """A caller-supplied objective as the generation loops use it: ``ext(ctx, X)`` -> (P,) device tensor."""

    def __init__(self, tagged, args):
        self.fun = tagged.fun
        self.args = tuple(args) if args not in ((), None) else ()
        self.name = tagged.__name__

    def __call__(self, ctx, X):
        t = __import__("torch")
        P = X.shape[0]
        f = self.fun(X, *self.args)
        if not isinstance(f, t.Tensor) or f.device != X.device:
            raise TypeError(f"batched objective {self.name}: expected a tensor on {X.device}, got {type(f).__name__}")
        if f.shape != (P,):
            raise ValueError(f"batched objective {self.name}: expected shape ({P},), got {tuple(f.shape)}")
        return f.to(t.float64).contiguous()


class HostPool:
    """Host workers for a plain Python objective (SURVEY.md section 8b case iii): what the reference's backend hook does
    with ``workers`` and ``backend in {"threading", "loky"}`` (_common.py:38-43, 94-97: a joblib pool of threads /
    processes, one ``fun(xx, *args)`` per individual).  Same calls on the same rows, so the same values as the serial
    loop; the differences are mechanical: rows travel in contiguous blocks (one task per block instead of one per
    individual: a 4096-row population is 4 x workers tasks, not 4096) and the pool outlives the generation.
    ``threading``: a ThreadPoolExecutor (pays off when ``fun`` releases the GIL -- numpy, I/O, an external solver);
    ``loky``: joblib's reusable process executor (``fun`` and ``args`` must pickle; cloudpickle takes lambdas / closures)."""

    BACKENDS = ("threading", "loky")

    def __init__(self, workers, backend=None):
        import os

        backend = backend if backend else "threading"  # the reference's default, _common.py:94
        if backend not in self.BACKENDS:
            raise ValueError(f"unknown host backend '{backend}' (expected one of {self.BACKENDS})")
        workers = int(workers)
        if workers < 0:  # joblib's convention: -1 = all cores, -2 = all but one
            workers = max(1, (os.cpu_count() or 1) + 1 + workers)
        if workers < 2:
            raise ValueError(f"host_workers={workers}: a pool needs at least two workers")
        self.workers, self.backend = workers, backend
        self._threads = None

    def executor(self):
        if self.backend == "loky":
            from joblib.externals.loky import get_reusable_executor

            return get_reusable_executor(max_workers=self.workers, timeout=600)
        if self._threads is None:
            from concurrent.futures import ThreadPoolExecutor

            self._threads = ThreadPoolExecutor(max_workers=self.workers, thread_name_prefix="sx-host")
        return self._threads

    def close(self):
        if self._threads is not None:
            self._threads.shutdown(wait=True)
            self._threads = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _eval_block(fun, block, args):
    """One pool task: the reference's serial wrapper (_common.py:79-80) on a block of rows."""
    return np.array([fun(xx, *args) for xx in block], dtype=np.float64)


class HostExternal(External):
    """Any other Python callable (SURVEY.md section 8b case iii): the CALLER'S scalar function, evaluated the way
    the reference's wrappers do it -- serially, ``np.array([fun(xx, *args) for xx in x])`` (_common.py:79-80), or, with
    a HostPool, by host threads / processes as its joblib backends do (_common.py:38-43) -- on a host copy of the
    candidates, between the propose and the select kernel.  Draws, mutation / velocity update, selection and
    bookkeeping stay on the device; what is slow is the caller's Python plus a D2H / H2D round trip per generation (such
    a generation cannot be captured into a graph).  With a pool the candidates come down in pieces (asynchronous copies
    into pinned memory, an event per piece) and the first pieces are being evaluated while the later ones still travel.
    It is the caller's function that runs on the host here, never this package's restatement of anything."""

    capturable = False
    _warned = False

    def __init__(self, fun, args, pool=None, from_reference_options=False):
        self.fun = fun
        self.args = tuple(args) if args not in ((), None) else ()
        self.name = getattr(fun, "__name__", "objective")
        self.pool = pool
        self.from_reference_options = from_reference_options  # workers / backend named the HOST pool (reference spelling)
        self._pinned = None

    def _check(self, f, P):
        if f.shape != (P,):
            raise ValueError(f"objective {self.name}: expected one scalar per individual, got shape {f.shape}")

    def __call__(self, ctx, X):
        t = __import__("torch")
        if not HostExternal._warned and self.pool is None:
            HostExternal._warned = True
            warnings.warn(
                f"stochopy_amd: {self.name} is a plain Python callable -- it is evaluated on the host, one call per "
                "individual, with a device-host round trip per generation.  Use stochopy_amd.factory objectives "
                "(fused kernels) or factory.batched (device tensor in/out) for speed, or options['host_workers'] "
                "(with 'host_backend' = 'threading' | 'loky') to spread the calls over host cores as the reference's "
                "joblib backends do.", RuntimeWarning, stacklevel=3)
        P = X.shape[0]
        if self.pool is None:
            ctx.sync()
            x = X.cpu().numpy()
            f = np.array([self.fun(xx, *self.args) for xx in x], dtype=np.float64)  # reference _common.py:79-80
            self._check(f, P)
            return t.from_numpy(np.ascontiguousarray(f)).to(X.device)
        return self._pooled(ctx, X, t)

    def _pooled(self, ctx, X, t):
        P, n = X.shape
        X = X.contiguous()
        if self._pinned is None or self._pinned.shape != (P, n):
            self._pinned = t.empty((P, n), dtype=t.float64).pin_memory()
        host = self._pinned
        ex = self.pool.executor()
        # tasks: 4 per worker (load balance against uneven objectives); copies: up to 8 pieces of whole tasks
        per_task = max(1, -(-P // (4 * self.pool.workers)))
        bounds_ = list(range(0, P, per_task)) + [P]
        ntask = len(bounds_) - 1
        pieces = min(8, ntask)
        cut = [bounds_[(ntask * j) // pieces] for j in range(pieces)] + [P]
        events = []
        with t.cuda.stream(ctx.stream):
            for j in range(pieces):
                host[cut[j]:cut[j + 1]].copy_(X[cut[j]:cut[j + 1]], non_blocking=True)
                ev = t.cuda.Event()
                ev.record(ctx.stream)
                events.append(ev)
        xh = host.numpy()
        # processes receive a pickled copy of their block; threads read the pinned buffer in place
        futures, k = [], 0
        for j in range(pieces):
            events[j].synchronize()
            while k < ntask and bounds_[k + 1] <= cut[j + 1]:
                futures.append(ex.submit(_eval_block, self.fun, xh[bounds_[k]:bounds_[k + 1]], self.args))
                k += 1
        parts = [fu.result() for fu in futures]
        for part, a, b in zip(parts, bounds_[:-1], bounds_[1:]):
            self._check(part, b - a)
        f = np.concatenate(parts)
        return t.from_numpy(np.ascontiguousarray(f)).to(X.device)


def resolve_objective(fun, args, workers=1, backend=None, host_workers=None, host_backend=None):
    """Map the user's callable to a device kernel id, or to an External for caller-supplied objectives -- the
    three cases of the backend hook contract (reference _common.py:27-106):

    (i)   the factory objectives (stochopy_amd.factory, tagged with ``sx_id``) run fused in the generation kernels;
    (ii)  ``factory.batched(fun)`` (device tensor in, device tensor out) runs between a propose and a select kernel;
    (iii) any other callable is the reference's ``fun(x, *args)`` on ONE individual: it runs on the host, per row
          (HostExternal) -- serially with a one-time warning naming the cost, or on a pool of host threads / processes:
          ``host_workers=N`` (``-1`` = all cores) with ``host_backend="threading"|"loky"``, or the REFERENCE'S OWN
          spelling ``workers=N, backend="threading"|"loky"`` (_common.py:94-97), which for such an objective names
          exactly what it names there (the run then uses one GPU).
    """
    from ..factory.benchmark import batched

    if not hasattr(fun, "__call__"):
        raise TypeError()
    plain = not isinstance(fun, (Objective, batched))
    if not plain and (host_workers not in (None, 0, 1) or host_backend is not None):
        raise ValueError("host_workers / host_backend apply to plain Python objectives (evaluated on the host) only")
    if isinstance(fun, Objective):
        if args not in ((), None):
            raise TypeError("factory objectives take no extra args")
        return fun.sx_id
    if isinstance(fun, batched):
        return External(fun, args)
    if backend in HostPool.BACKENDS:  # the reference's spelling
        if host_workers is not None or host_backend is not None:
            raise ValueError("give either workers / backend in the reference's spelling or host_workers / host_backend")
        if workers in (None, 0, 1):  # the reference runs serially then (_common.py:96)
            return HostExternal(fun, args, None, True)
        return HostExternal(fun, args, HostPool(workers, backend), True)
    if host_workers in (None, 0, 1):
        if host_backend is not None and host_backend not in HostPool.BACKENDS:
            raise ValueError(f"unknown host backend '{host_backend}' (expected one of {HostPool.BACKENDS})")
        return HostExternal(fun, args)
    return HostExternal(fun, args, HostPool(host_workers, host_backend))


def resolve_updating(updating, strict_updating, workers, fun, ndim=0):
    """Is the run an ordered sweep (the reference's default ``updating="immediate"``: de_async / pso_async)?

    The reference runs it whenever no parallel backend is chosen (de/_de.py:142-145).  Here the sweep is one
    kernel per generation on ONE GPU with the objective inside it, so it needs ``workers == 1`` and a factory
    objective.  ``strict_updating``: None (default) honours "immediate" when that is possible and otherwise
    switches to "deferred" WITH a warning; True insists (an impossible combination raises); False always defers,
    silently -- the throughput choice, like picking a parallel backend in the reference."""
    if updating != "immediate":
        return False
    import os

    # (SX_FORCE_SHARDED=1: the test switch that runs a 1-rank process group through the sharded path)
    # (the sweep keeps the row of the individual at work in one workgroup's LDS: rows of up to _lib.NARROW_DIM elements)
    from .._lib import NARROW_DIM

    possible = (workers == 1 and isinstance(fun, int) and os.environ.get("SX_FORCE_SHARDED") != "1"
                and ndim <= NARROW_DIM)
    if strict_updating is None:
        if not possible:
            warnings.warn('stochopy_amd: updating="immediate" is an ordered sweep on one GPU with a fused factory '
                          'objective and rows of <= 4096 elements; this run (workers > 1, a caller-supplied objective or longer rows) uses "deferred" updating, '
                          "as a parallel backend of the reference does (de/_de.py:142-145).", RuntimeWarning, stacklevel=3)
        if possible:
            _note_serial_sweep()
        return possible
    if strict_updating and not possible:
        raise ValueError('strict_updating=True: updating="immediate" is an ordered sweep on ONE GPU with a fused factory '
                         "objective and rows of <= 4096 elements; this call (workers > 1, a caller-supplied objective, "
                         "longer rows or a sharded process group) cannot run it")
    return bool(strict_updating)


_sweep_noted = False


def _note_serial_sweep():
    """Once per process: a DEFAULT call (updating="immediate", the reference's default) runs the ordered sweep -- the
    reference's result for the same seed, on one workgroup -- which is orders of magnitude slower than the
    deferred generation kernels.  Say so, so that nobody benchmarks the parity mode by accident."""
    global _sweep_noted
    if not _sweep_noted:
        _sweep_noted = True
        warnings.warn('stochopy_amd: updating="immediate" (the default, as in the reference) runs one ORDERED sweep per '
                      'generation on a single workgroup -- the reference\'s semantics and, for the same seed, its result.  '
                      'For throughput pass updating="deferred" (or strict_updating=False): whole-population generation '
                      "kernels, 100x faster at large popsize.  (Shown once.)", UserWarning, stacklevel=4)


def evaluate_rows(ctx, fun, X, n, f, xm=None, xstd=None, clip=False):
    """f[i] = fun(row i of X) -- optionally of clip(X, -1, 1) (Penalize) and of X * xstd + xm (CMA-ES
    standardisation, cmaes/_cmaes.py:167-173) -- by the fused device kernel or the caller's objective."""
    from .. import _device, _lib

    if isinstance(fun, int):
        if not clip:
            return _device.evaluate(ctx, fun, X, n, f=f, xm=xm, xstd=xstd)
        p = _device.ptr
        _lib.check(ctx.L.sx_cmaes_eval_penalized(fun, p(X), X.shape[0], n, p(xm), p(xstd), None, p(f), None,
                                                 ctx.stream_ptr), "sx_cmaes_eval_penalized")
        return f
    Xs = X.clamp(-1.0, 1.0) if clip else X
    f.copy_(fun(ctx, Xs if xm is None else Xs * xstd + xm))
    return f


def penalty_rows(ctx, fun, X, n, xm, xstd, v, f_raw, pen):
    """Penalize's second pass (cmaes/_constraints.py:79): pen[i] = sum_j (clip(x_ij) - x_ij)^2 * v[j]."""
    from .. import _device, _lib

    if isinstance(fun, int):
        p = _device.ptr
        _lib.check(ctx.L.sx_cmaes_eval_penalized(fun, p(X), X.shape[0], n, p(xm), p(xstd), p(v), p(f_raw), p(pen),
                                                 ctx.stream_ptr), "sx_cmaes_eval_penalized")
    else:
        pen.copy_((((X.clamp(-1.0, 1.0) - X) ** 2) * v).sum(dim=1))


def resolve_workers(workers, fun_id=None):
    """`workers` = number of GPUs (processes are launched by torchrun; see parallel.py) -- unless the call named a host
    pool in the reference's spelling (plain Python objective with backend="threading"|"loky"): one GPU then."""
    if workers in (None, 0, 1) or getattr(fun_id, "from_reference_options", False):
        return 1
    if int(workers) == -1:  # "all": every rank of the initialised process group (one GPU without one)
        try:
            import torch.distributed as dist

            return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        except ImportError:
            return 1
    if int(workers) < 1:
        raise ValueError(f"workers={workers}: expected the number of GPUs (or -1 for all)")
    return int(workers)


_warned_replicated = set()


def replicated_workers(method, workers, why):
    """`workers > 1` where the run cannot be sharded (the reference's own random stream for DE / PSO / CPSO: one host
    stream, drawn in the order of the whole population; NA: every walk reads the whole model store): every rank -- or the one
    process there is -- carries the WHOLE run on its GPU.  The reference's invariant holds as it does for its parallel
    backends (tests/helpers.py:28-36: the backend must not change the result for a seed); there is nothing to gain either:
    such a run is bound by the host stream / one workgroup row per sample, not by the objective evaluations the reference
    hands to its workers (_common.py:58-72).  Says so once per method.  Returns 1."""
    if method not in _warned_replicated:
        _warned_replicated.add(method)
        warnings.warn(f"stochopy_amd: {method} with workers={workers} runs replicated -- {why}: same result as workers=1 on "
                      "every rank, no speed-up.  (Shown once.)", UserWarning, stacklevel=4)
    return 1


def resolve_rng(rng):
    rng = "numpy-legacy" if rng is None else rng
    if rng not in RNG_MODES:
        raise ValueError(f"rng must be one of {RNG_MODES}")
    return rng


def as_bounds(bounds):
    if np.ndim(bounds) != 2:
        raise ValueError()
    lower, upper = np.transpose(np.asarray(bounds, dtype=np.float64))
    return np.ascontiguousarray(lower), np.ascontiguousarray(upper)


def host_blas_single_thread():
    """Context manager: host BLAS calls inside run on one thread (threadpoolctl, when it is installed; a no-op otherwise).
    The host side of a device-resident run touches a few n-vectors; a threaded BLAS answers a 16 384-element dot product by
    waking every one of its spinning worker threads, which in a CPU-quota'd container (the GPU boxes here: 16 CPUs' quota on
    a 256-CPU host) gets the process throttled for tens of milliseconds -- longer than 100 generations of device work
    (profiles/r5_host_blas_throttle.txt)."""
    try:
        from threadpoolctl import threadpool_limits

        return threadpool_limits(limits=1, user_api="blas")
    except Exception:  # noqa: BLE001  (not installed, or a BLAS it does not know)
        import contextlib

        return contextlib.nullcontext()
