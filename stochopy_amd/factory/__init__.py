"""Objective handles for the MI355X backend: the seven benchmark functions as device kernels, plus the two tags
(`batched`) under which a caller's own device objective is accepted (see factory/benchmark.py)."""
from . import benchmark as _benchmark
from .benchmark import Objective  # noqa: F401

__all__ = list(_benchmark.__all__)
globals().update({_name: getattr(_benchmark, _name) for _name in __all__})
