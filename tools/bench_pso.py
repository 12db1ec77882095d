"""PSO generation cost across objectives / shapes (is the kernel VALU- or HBM-bound?)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


def per_gen(make, short=100, long=1100):
    wall(lambda: make(short))
    t1, r1 = wall(lambda: make(short)); t2, r2 = wall(lambda: make(long))
    return (t2 - t1) / (r2.nit - r1.nit)


cases = sys.argv[1:] or ["ackley:256:16384", "sphere:256:16384", "rosenbrock:256:16384", "rastrigin:256:16384",
                         "sphere:1024:16384", "ackley:1024:16384", "sphere:128:16384", "sphere:64:32768"]
for c in cases:
    name, n, P = c.split(":")
    n, P = int(n), int(P)
    o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}
    per = per_gen(lambda m: sa.optimize.minimize(getattr(sa.factory, name), [[-5.12, 5.12]] * n, method="pso",
                                                 options=dict(o, maxiter=m)))
    byts = (48 * n + 24) * P
    print(f"pso {name:12s} n={n:5d} P={P:6d}: {per*1e6:8.1f} us/gen  {byts/per/1e9:8.1f} GB/s (generation incl. finalise)")
