"""updating="immediate" (sequential sweeps, csrc/sx_async.hip): time per generation at the metric shapes."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


cases = sys.argv[1:] or ["de:rosenbrock:128:4096", "de:rosenbrock:2:8", "pso:ackley:256:16384", "cpso:ackley:256:16384",
                         "de:rosenbrock:1024:4096"]
for c in cases:
    method, name, n, P = c.split(":")
    n, P = int(n), int(P)
    o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "backend": "hip", "strict_updating": True}
    run = lambda m: sa.optimize.minimize(getattr(sa.factory, name), [[-5.12, 5.12]] * n, method=method,
                                         options=dict(o, maxiter=m))
    wall(lambda: run(3))
    t1, r1 = wall(lambda: run(3)); t2, r2 = wall(lambda: run(11)); t3, r3 = wall(lambda: run(43))
    per = (t2 - t1) / (r2.nit - r1.nit); late = (t3 - t2) / (r3.nit - r2.nit)
    print(f"immediate {method:4s} {name:10s} n={n:5d} P={P:6d}: {per*1e3:9.3f} ms/generation = {per/P*1e6:7.3f} us/individual, "
          f"{P/per:10.3e} evals/s; generations 12-43: {late*1e3:8.3f} ms")
