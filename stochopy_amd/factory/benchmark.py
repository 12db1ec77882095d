"""The seven benchmark objectives, as device-kernel handles.

Reference: stochopy/factory/benchmark.py:14-156 (same names, same values).
Each object is callable like the reference function -- ``rosenbrock(x)`` with a
1-D array returns a float, with a 2-D (P, n) array returns (P,) values -- but the
arithmetic runs in the HIP kernel ``sx_eval`` on the GPU (no CPU fallback).
``minimize(..., options={"backend": "hip"})`` recognises these handles and fuses
the objective into the generation kernels.
"""
import numpy as np

from .. import _lib

__all__ = [
    "ackley",
    "griewank",
    "quartic",
    "rastrigin",
    "rosenbrock",
    "sphere",
    "styblinski_tang",
    "batched",
]


class Objective:
    """Handle of a device-resident objective (``sx_id`` = SX_FUN_* of include/stochopy_hip.h)."""

    def __init__(self, name):
        self.__name__ = name
        self.name = name
        self.sx_id = _lib.FUN_IDS[name]

    def __repr__(self):
        return f"<stochopy_amd objective {self.name} (HIP kernel id {self.sx_id})>"

    def __call__(self, x):
        from .. import _device

        ctx = _default_context()
        x = np.asarray(x, dtype=np.float64)
        single = x.ndim == 1
        X = np.ascontiguousarray(x[None, :] if single else x)
        dX = ctx.upload(X)
        f = _device.evaluate(ctx, self.sx_id, dX, X.shape[1])
        ctx.sync()
        out = f.cpu().numpy()
        return float(out[0]) if single else out


_ctx = None


def _default_context():
    global _ctx
    if _ctx is None:
        from .. import _device

        _ctx = _device.Context()
    return _ctx


ackley = Objective("ackley")
griewank = Objective("griewank")
quartic = Objective("quartic")
rastrigin = Objective("rastrigin")
rosenbrock = Objective("rosenbrock")
sphere = Objective("sphere")
styblinski_tang = Objective("styblinski_tang")


class batched:
    """Tag a user objective that works on the device population: ``fun(X, *args)`` receives the (P, n) float64
    ROCm tensor of candidates (on the engine's current stream) and returns the (P,) float64 fitness tensor on the
    same device.  This is the backend's hook contract (reference _common.py:27-106: after decoration ``fun(X)``
    maps the (P, n) population to (P,) values).  The objective cannot be fused into the generation kernels, so a
    generation becomes propose -> fun -> select (csrc/sx_unfused.hip); everything else stays as it is.

    This package itself never evaluates an objective on the host.  A caller whose objective only exists as
    numpy code can do the round trip inside ``fun`` (``X.cpu().numpy()`` in, a tensor on ``X.device`` out) -- it is
    then their code that is slow, visibly, and the generations are launched one by one instead of as a graph."""

    def __init__(self, fun):
        if not hasattr(fun, "__call__"):
            raise TypeError()
        self.fun = fun
        self.__name__ = getattr(fun, "__name__", "objective")

    def __repr__(self):
        return f"<stochopy_amd device-batched objective {self.__name__}>"

    def __call__(self, *a, **k):
        return self.fun(*a, **k)
