"""Public surface of the MI355X backend's optimizers.

The names a stochopy user expects under ``stochopy.optimize`` (reference optimize/__init__.py:1-18) resolve
here to the HIP-backed front ends (all six of the reference's optimizers).
Importing a front end also registers it with ``minimize(method=...)``.
"""
import importlib

from . import _helpers

OptimizeResult = _helpers.OptimizeResult
minimize = _helpers.minimize
register = _helpers.register

# method name -> module holding its `minimize` front end (each module registers itself on import)
_FRONT_ENDS = {"de": "_de", "pso": "_pso", "cpso": "_cpso", "cmaes": "_cmaes", "vdcma": "_vdcma", "na": "_na"}
for _method, _module in _FRONT_ENDS.items():
    globals()[_method] = importlib.import_module("." + _module, __name__).minimize

__all__ = ["OptimizeResult", "minimize", "register", *_FRONT_ENDS]
