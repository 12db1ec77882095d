"""Same public names as stochopy.optimize (reference optimize/__init__.py:1-18),
restricted to the methods on the MI355X hot path."""
from ._helpers import OptimizeResult, minimize, register
from ._cmaes import minimize as cmaes
from ._cpso import minimize as cpso
from ._de import minimize as de
from ._pso import minimize as pso
from ._vdcma import minimize as vdcma

__all__ = ["OptimizeResult", "minimize", "register", "cmaes", "cpso", "de", "pso", "vdcma"]
