"""PSO front end: CPSO without the competitive restart.

Reference: stochopy/optimize/pso/_pso.py:9-122 (same signature minus ``competitivity``;
forwards to cpso.minimize with ``competitivity=None``).
"""
from . import _cpso
from ._helpers import register

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
    """Minimize an objective function using PSO on MI355X (reference pso/_pso.py:9-29)."""
    return _cpso.minimize(fun, bounds, x0, args, maxiter, popsize, inertia, cognitivity, sociability, None, seed,
                          xtol, ftol, constraints, updating, workers, backend, return_all, verbosity, callback, rng,
                          strict_updating, host_workers, host_backend)


register("pso", minimize)
