"""Shared pieces of the per-method front ends.

Reference: stochopy/optimize/_common.py:13-24 (messages), :27-106 (the backend
hook this package sits beside).  Where the reference's hook wraps the scalar
objective into a population-level callable and parallelises only those calls,
``backend="hip"`` moves the whole generation onto the GPU; the hook contract that
is kept is the user-visible one: same ``minimize`` options, synchronous updating
whenever a parallel backend is chosen (de/_de.py:142-145), identical results for
a given seed (rng="numpy-legacy").
"""
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


def resolve_backend(backend):
    """`backend="hip"` (or None: this package has only that one)."""
    if backend in (None, "hip"):
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


def resolve_objective(fun, args):
    """Map the user's callable to a device kernel id, or to an External for tagged caller-supplied objectives.

    The factory objectives (stochopy_amd.factory, tagged with ``sx_id``) run fused in the generation kernels.
    ``factory.batched(fun)`` (device tensor in, device tensor out) runs between a propose and a select kernel.
    Untagged callables are refused: this package never evaluates an objective on the host.
    """
    from ..factory.benchmark import batched

    if not hasattr(fun, "__call__"):
        raise TypeError()
    if isinstance(fun, Objective):
        if args not in ((), None):
            raise TypeError("factory objectives take no extra args")
        return fun.sx_id
    if isinstance(fun, batched):
        return External(fun, args)
    raise TypeError(
        "backend='hip' needs a device objective from stochopy_amd.factory "
        "(ackley, griewank, quartic, rastrigin, rosenbrock, sphere, styblinski_tang) or a callable that works on "
        f"the device population, tagged with stochopy_amd.factory.batched; got {fun!r}.  There is no host fallback.")


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


def resolve_workers(workers):
    """`workers` = number of GPUs (processes are launched by torchrun; see parallel.py)."""
    if workers in (None, 0, 1):
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
