"""Same public names as stochopy.optimize (reference optimize/__init__.py:1-18),
restricted to the methods on the MI355X hot path."""
from ._helpers import OptimizeResult, minimize, register
from ._de import minimize as de

__all__ = ["OptimizeResult", "minimize", "register", "de"]
