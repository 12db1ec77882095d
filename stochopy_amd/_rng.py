"""Random draws for the generation loops (product side).

``LegacyHostStream`` drives the C++ MT19937 replica (csrc/sx_mt19937.cpp) and is
interchangeable with numpy's legacy *global* generator: it starts from
``np.random.get_state()`` (after ``np.random.seed(seed)`` when a seed is given,
exactly what the reference does at de/_de.py:148-149) and ``sync_back()`` stores
the advanced state into numpy again, so a run consumes the global stream exactly
like the reference would.  Draw order per generation: SURVEY.md Appendix B.
"""
import ctypes as C

import numpy as np

from . import _lib


class LegacyHostStream:
    kind = "numpy-legacy"

    def __init__(self, seed=None):
        L = _lib.lib()
        self._L = L
        if seed is not None:
            np.random.seed(seed)  # reference: `if seed is not None: np.random.seed(seed)`
        name, key, pos, has_gauss, gauss = np.random.get_state()
        assert name == "MT19937"
        self._h = L.sx_mt_create(0)
        key = np.ascontiguousarray(key, dtype=np.uint32)
        L.sx_mt_set_state(self._h, key.ctypes.data, int(pos), int(has_gauss), float(gauss))

    def __del__(self):
        try:
            if self._h:
                self._L.sx_mt_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def sync_back(self):
        """Write the advanced state into numpy's global generator."""
        key = np.empty(624, dtype=np.uint32)
        pos = C.c_int()
        hg = C.c_int()
        g = C.c_double()
        self._L.sx_mt_get_state(self._h, key.ctypes.data, C.byref(pos), C.byref(hg), C.byref(g))
        np.random.set_state(("MT19937", key, pos.value, hg.value, g.value))

    # -- primitives -----------------------------------------------------------
    def random(self, shape, out=None):
        out = np.empty(shape) if out is None else out
        self._L.sx_mt_random(self._h, out.ctypes.data, out.size)
        return out

    def uniform_rows(self, lower, upper, rows, out=None):
        lower = np.ascontiguousarray(lower, dtype=np.float64)
        upper = np.ascontiguousarray(upper, dtype=np.float64)
        n = lower.size
        out = np.empty((rows, n)) if out is None else out
        self._L.sx_mt_uniform_rows(self._h, lower.ctypes.data, upper.ctypes.data, n, rows, out.ctypes.data)
        return out

    def uniform(self, lo, hi, count):
        out = np.empty(count)
        self._L.sx_mt_uniform(self._h, float(lo), float(hi), out.ctypes.data, count)
        return out

    def randn(self, shape, out=None):
        out = np.empty(shape) if out is None else out
        self._L.sx_mt_randn(self._h, out.ctypes.data, out.size)
        return out

    def randint(self, high, count):
        out = np.empty(count, dtype=np.int64)
        self._L.sx_mt_randint(self._h, int(high), out.ctypes.data, count)
        return out

    def permutation(self, n):
        out = np.empty(n, dtype=np.int64)
        self._L.sx_mt_permutation(self._h, n, out.ctypes.data)
        return out

    def de_donors(self, P, k, out=None):
        out = np.empty((k, P), dtype=np.int32) if out is None else out
        self._L.sx_mt_de_donors(self._h, P, k, out.ctypes.data)
        return out

    def de_async_draws(self, P, k, n, donors, irand, lower=None, upper=None, resample=None):
        """The per-individual draws of one de_async generation after r1 (de/_de.py:376-382), into the arrays."""
        lo = hi = rs = None
        if resample is not None:
            lower = np.ascontiguousarray(lower, dtype=np.float64)
            upper = np.ascontiguousarray(upper, dtype=np.float64)
            lo, hi, rs = lower.ctypes.data, upper.ctypes.data, resample.ctypes.data
        self._L.sx_mt_de_async_draws(self._h, P, k, n, donors.ctypes.data, irand.ctypes.data, lo, hi, rs)

    # -- composite: initial population (reference _common.py:109-120) ----------
    def latin_hypercube(self, P, n, lower, upper):
        # one pass in the stream's own library (the strided column gathers of the numpy form cost 3 ms at P = 4096,
        # n = 128 on the GPU box's host -- a fifth of a 1000-generation run); same draws, same operations, same bits
        lin = np.linspace(-1.0, 1.0, P, endpoint=False)
        scale = np.ascontiguousarray(np.broadcast_to(0.5 * (upper - lower), (n,)), dtype=np.float64)
        shift = np.ascontiguousarray(np.broadcast_to(0.5 * (upper + lower), (n,)), dtype=np.float64)
        pop = np.empty((P, n))
        self._L.sx_mt_latin_hypercube(self._h, P, n, lin.ctypes.data, scale.ctypes.data, shift.ctypes.data,
                                      pop.ctypes.data)
        return pop


def philox_key(seed):
    """(key0, key1) of the in-kernel Philox4x32-10 generator."""
    if seed is None:
        raise ValueError('rng="philox" needs an explicit integer seed')
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return seed & 0xFFFFFFFF, seed >> 32


def make_init_stream(rng, seed):
    """Stream used for the host-side initial population / initial mean.

    numpy-legacy: the global stream (seeded if a seed is given).
    philox: a private legacy stream seeded with the low 32 bits of the seed
    (same rule as oracle/streams.py PhiloxStream) -- numpy's global state is untouched.
    """
    if rng == "numpy-legacy":
        return LegacyHostStream(seed)
    philox_key(seed)  # rng="philox" without a seed: the explanatory ValueError, not an int(None) TypeError
    state = np.random.get_state()
    try:
        return LegacyHostStream(int(seed) & 0xFFFFFFFF)
    finally:
        np.random.set_state(state)


def philox_latin_hypercube(ctx, out, row0, Ptotal, d_lower, d_upper, seed):
    """Rows [row0, row0 + len(out)) of the Philox-mode initial population, drawn ON THE DEVICE straight into ``out``
    ((rows, n) device tensor): the reference's Latin hypercube (_common.py:109-120) with counter-based draws
    (csrc/sx_core.hip philox_lhs_kernel; oracle counterpart oracle/streams.py PhiloxStream.lhs_population).  Nothing is
    built on the host, and a rank of a sharded run draws only its own rows."""
    from . import _device

    key0, key1 = philox_key(seed)
    rows, n = out.shape
    _lib.check(ctx.L.sx_philox_lhs(_device.ptr(out), rows, n, out.stride(0), int(row0), int(Ptotal), _device.ptr(d_lower),
                                   _device.ptr(d_upper), key0, key1, ctx.stream_ptr), "sx_philox_lhs")
    return out
