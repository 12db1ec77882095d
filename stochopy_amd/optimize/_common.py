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


def resolve_objective(fun, args):
    """Map the user's callable to a device kernel id.

    Only the factory objectives (stochopy_amd.factory, tagged with ``sx_id``) run
    fused on the device.  Arbitrary Python callables would have to be evaluated
    on the host every generation -- that is the reference's own CPU path, not
    this backend's, so it is refused instead of silently falling back.
    """
    if not hasattr(fun, "__call__"):
        raise TypeError()
    if isinstance(fun, Objective):
        if args not in ((), None):
            raise TypeError("factory objectives take no extra args")
        return fun.sx_id
    raise TypeError(
        "backend='hip' needs a device objective from stochopy_amd.factory "
        "(ackley, griewank, quartic, rastrigin, rosenbrock, sphere, styblinski_tang); "
        f"got {fun!r}.  There is no host fallback.")


def resolve_workers(workers):
    """`workers` = number of GPUs (processes are launched by torchrun; see parallel.py)."""
    if workers in (None, 0, 1):
        return 1
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
