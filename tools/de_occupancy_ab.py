"""DE best1bin generations over population sizes from the metric's 4096 to 2^20 rows (one-batch and off-grid row lengths): us per
generation and the fraction of the HBM peak on (32 n + 16) B per evaluation.  Run once per build (tools/ab_lib.py <lib.so> this)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
from stochopy_amd import _lib
print("library:", _lib.LIB_PATH, flush=True)


def per_gen(fun, n, P, short, long_, reps=2):
    o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, updating="deferred", strategy="best1bin")

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(fun, [[-5.12, 5.12]] * n, method="de", options=dict(o, maxiter=m))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r.nit
    wall(short)
    runs = [(wall(short), wall(long_)) for _ in range(reps)]
    (t1, n1), (t2, n2) = min(r[0] for r in runs), min(r[1] for r in runs)
    return (t2 - t1) / (n2 - n1)


for name, n, P in (("rosenbrock", 128, 4096), ("rastrigin", 128, 4096), ("rosenbrock", 128, 16384), ("rosenbrock", 128, 65536),
                   ("rosenbrock", 128, 1 << 20), ("rastrigin", 128, 1 << 20), ("rosenbrock", 64, 1 << 20), ("rosenbrock", 256, 1 << 19),
                   ("ackley", 256, 1 << 17), ("rosenbrock", 100, 65536), ("rosenbrock", 200, 65536), ("rosenbrock", 300, 65536),
                   ("rosenbrock", 1000, 16384)):
    gens = max(60, min(3000, int(0.15 / (4 * 8 * n * P / 3e12 + 3e-6))))
    t = per_gen(getattr(sa.factory, name), n, P, max(10, gens // 6), gens)
    print(f"DE best1bin {name:10s} n={n:5d} P={P:8d}: {t*1e6:9.2f} us/generation  {(32*n+16)*P/t/1e9/8000:.3f} of 8 TB/s", flush=True)
