"""Where the fixed cost of a whole minimize() call at M goes: cProfile of a 2-generation call (after warm-up calls)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa
b = [[-5.12, 5.12]] * 128
o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred", "strategy": "best1bin"}
def wall(m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(o, maxiter=m))
    torch.cuda.synchronize(); return time.perf_counter() - t0, r
for _ in range(3): wall(50)
for m in (2, 1000):
    print("maxiter", m, "min of 5: %.3f ms" % (min(wall(m)[0] for _ in range(5)) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): wall(2)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
