"""CPSO generation cost while the competitive restart fires every generation (maxiter <= 122 at P = 16384)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa

b = [[-5.12, 5.12]] * 256
o = {"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}


def wall(m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sa.optimize.minimize(sa.factory.ackley, b, method="cpso", options=dict(o, maxiter=m))
    torch.cuda.synchronize(); return time.perf_counter() - t0, r


wall(20)
t1, r1 = wall(20); t2, r2 = wall(120)
print("C3b cpso ackley n256 P16384, restart firing every generation: %.1f us/gen" % ((t2 - t1) / (r2.nit - r1.nit) * 1e6))
