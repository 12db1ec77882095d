"""What a whole minimize() call costs besides its generations: wall of 2-generation calls (after warm-up), per method."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa
warnings.simplefilter("ignore")

def wall(method, n, P, extra, m):
    b = [[-5.12, 5.12]] * n
    o = dict({"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": m}, **extra)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sa.optimize.minimize(sa.factory.rosenbrock, b, method=method, options=o)
    torch.cuda.synchronize(); return time.perf_counter() - t0

CASES = [("de", 128, 4096, {"updating": "deferred"}), ("de", 128, 4096, {"updating": "immediate"}), ("pso", 256, 16384, {"updating": "deferred"}),
         ("cpso", 256, 16384, {"updating": "deferred"}), ("cmaes", 10, 20, {}), ("cmaes", 20, 40, {}), ("cmaes", 32, 64, {}), ("cmaes", 64, 128, {}), ("cmaes", 512, 1024, {}), ("vdcma", 512, 64, {}),
         ("na", 8, 64, {})]
for method, n, P, extra in CASES:
    for _ in range(3): wall(method, n, P, extra, 3)
    t2 = min(wall(method, n, P, extra, 2) for _ in range(7))
    t12 = min(wall(method, n, P, extra, 42) for _ in range(7))
    per = (t12 - t2) / 40
    print("%-6s n=%4d P=%6d %-26s 2 generations: %7.3f ms   (per generation %8.2f us -> fixed part ~%6.3f ms)" % (method, n, P, extra, t2 * 1e3, per * 1e6, (t2 - 2 * per) * 1e3), flush=True)
