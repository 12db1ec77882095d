import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
b = [[-5.12, 5.12]] * 128
o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred", "strategy": "best1bin"}
def wall(m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(o, maxiter=m))
    torch.cuda.synchronize(); return time.perf_counter() - t0, r
wall(50)
for m in (1, 2, 64, 512, 2000, 8000):
    ts = [wall(m)[0] for _ in range(5)]
    print(f"maxiter {m:5d}: min {min(ts)*1e3:8.3f} ms  median {sorted(ts)[2]*1e3:8.3f} ms  -> {min(ts)/m*1e6:8.2f} us/gen")
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); wall(2000); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
