"""Round 4 experiment: does the kernel boundary at the metric shape wait for the write-back of the generation's 4 MB of
row stores?  Per generation (two run lengths, best of three) for the library given through tools/ab_lib.py; with
SX_POP_FLAGS=<n> the two population buffers come from hipExtMallocWithFlags(flags=n) (1 fine-grained, 3 uncached)
instead of torch's allocator.  usage: python tools/ab_lib.py <lib.so> tools/m_store_ab.py"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa
from stochopy_amd import _device

flags = int(os.environ.get("SX_POP_FLAGS", "0"))
if flags:
    hip = C.CDLL("libamdhip64.so")
    hip.hipExtMallocWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
    keep = []
    orig = _device.Context.empty

    def empty(self, shape, dtype=None):
        if isinstance(shape, tuple) and len(shape) == 2 and shape[0] >= 1024 and dtype is None:
            p = C.c_void_p()
            assert hip.hipExtMallocWithFlags(C.byref(p), shape[0] * shape[1] * 8, flags) == 0

            class _Mem:
                __cuda_array_interface__ = {"shape": shape, "typestr": "<f8", "data": (p.value, False), "version": 2, "strides": None}
            m = _Mem(); keep.append(m)
            return torch.as_tensor(m, device=self.device)
        return orig(self, shape, dtype)
    _device.Context.empty = empty


def per_gen(obj, short=1000, long_=9000):
    b = [[-5.12, 5.12]] * 128
    o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred", "strategy": "best1bin"}

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(getattr(sa.factory, obj), b, method="de", options=dict(o, maxiter=m))
        torch.cuda.synchronize(); return time.perf_counter() - t0, r

    wall(short)
    best = None
    for _ in range(3):
        (t1, r1), (t2, r2) = wall(short), wall(long_)
        v = (t2 - t1) / (r2.nit - r1.nit)
        best = v if best is None or v < best else best
    return best, r2.fun


for obj in ("rosenbrock", "rastrigin"):
    t, f = per_gen(obj)
    print(f"{os.path.basename(sa._lib.LIB_PATH):24s} flags={flags} DE {obj:10s} n128 P4096: {t*1e6:6.3f} us/gen -> {4096/t:.3e} evals/s (fun {f!r})", flush=True)
