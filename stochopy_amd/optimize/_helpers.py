"""API surface kept verbatim from the reference.

Reference: stochopy/optimize/_helpers.py:8-94 (``OptimizeResult``, ``register``,
``minimize``) and stochopy/_common.py:1-35 (``BaseResult``: dict with attribute
access whose repr sorts keys, right-justifies them and hides xall / funall).
"""

__all__ = ["OptimizeResult", "minimize", "register"]

_optimizer_map = {}


class OptimizeResult(dict):
    """Optimization result: keys x, success, status, message, fun, nfev, nit (+ xall, funall)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    __setattr__ = dict.__setitem__
    __delattr__ = dict.__delitem__

    def __repr__(self):
        if not self:
            return self.__class__.__name__ + "()"
        width = max(len(k) for k in self) + 1
        rows = [k.rjust(width) + ": " + repr(self[k]) for k in sorted(self) if k not in ("xall", "funall")]
        return "\n".join(rows)

    def __dir__(self):
        return list(self.keys())


def register(name, minimize):
    """Register an optimizer under ``method=name`` (reference _helpers.py:39-41)."""
    _optimizer_map[name] = minimize


def minimize(fun, bounds, x0=None, args=(), method="de", options=None, callback=None):
    """Minimize ``fun`` with a stochastic optimizer on the GPU.

    Same signature and dispatch as the reference (``_helpers.py:44-94``):
    ``options`` is splatted into the per-method function.  Methods offered:
    ``"de"``, ``"pso"``, ``"cpso"``, ``"cmaes"``, ``"vdcma"``, ``"na"``.  Options added by this backend:
    ``backend="hip"`` (the default here), ``workers`` = number of GPUs (-1 = every rank of the process group),
    ``rng`` in {"numpy-legacy", "philox"}, ``strict_updating`` (insist on / opt out of the ordered sweep of
    ``updating="immediate"``), and for DE with several GPUs ``exchange`` / ``donors``.  ``fun`` is a
    ``stochopy_amd.factory`` objective (fused kernels), a caller's own device objective tagged with
    ``factory.batched``, or any other Python callable ``fun(x, *args)`` (evaluated per individual on the host).
    """
    options = options if options else {}
    try:
        opt = _optimizer_map[method]
    except KeyError:
        raise KeyError(
            f"method '{method}' is not on the MI355X hot path (available: {sorted(_optimizer_map)}); "
            "use keurfonluu/stochopy itself for it") from None
    return opt(fun=fun, bounds=bounds, x0=x0, args=args, callback=callback, **options)
