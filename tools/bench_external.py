"""What the caller-supplied-objective path costs per generation (propose -> fun -> select -> finalise) next to
the fused kernels, at the metric shape."""
import ctypes as C, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
from stochopy_amd import _lib

L = _lib.lib()


def device_fun(name):
    fid = _lib.FUN_IDS[name]

    def fun(X):
        P, n = X.shape
        f = torch.empty((P,), dtype=torch.float64, device=X.device)
        L.sx_eval(fid, X.data_ptr(), P, n, n, None, None, f.data_ptr(), None, None,
                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return f
    return sa.factory.batched(fun)


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for method, name, n, P in (("de", "rosenbrock", 128, 4096), ("pso", "ackley", 256, 16384), ("de", "rosenbrock", 1024, 16384)):
    o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "backend": "hip", "updating": "deferred"}
    objs = {"fused": getattr(sa.factory, name), "sx_eval as a batched objective": device_fun(name),
            "torch rosenbrock/sum": sa.factory.batched(
                lambda X: (100.0 * (X[:, 1:] - X[:, :-1] ** 2) ** 2).sum(1) + ((1.0 - X[:, :-1]) ** 2).sum(1))}
    for label, f in objs.items():
        run = lambda m: sa.optimize.minimize(f, [[-5.12, 5.12]] * n, method=method, options=dict(o, maxiter=m))
        wall(lambda: run(100))
        t1, r1 = wall(lambda: run(100)); t2, r2 = wall(lambda: run(1100))
        print(f"{method} {name} n={n} P={P} {label:32s}: {(t2 - t1) / (r2.nit - r1.nit) * 1e6:8.1f} us/generation")
