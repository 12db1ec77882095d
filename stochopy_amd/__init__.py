"""stochopy_amd -- MI355X-native population-evaluation engine behind stochopy's API.

Mirrors the surface of keurfonluu/stochopy's optimisation hot path:
``stochopy_amd.optimize.minimize(fun, bounds, x0, args, method, options, callback)``
returns an ``OptimizeResult`` (reference: stochopy/optimize/_helpers.py:44-94);
``stochopy_amd.factory`` holds the seven benchmark objectives
(reference: stochopy/factory/benchmark.py), each tagged with its device kernel.

Every generation runs as hand-written HIP kernels (stochopy_amd/csrc) through
the C ABI of include/stochopy_hip.h.  There is no CPU fallback.
"""
__version__ = "0.1.0"

from . import factory, optimize  # noqa: F401,E402
