"""BASELINE configs 3a / 3b (PSO / CPSO, Ackley n=256, P=16384, Philox draws): evals/s from two run lengths of whole
minimize() calls (100 and 1100 generations), best of three."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa

b = [[-5.12, 5.12]] * 256
o = {"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}


def wall(method, m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sa.optimize.minimize(sa.factory.ackley, b, method=method, options=dict(o, maxiter=m))
    torch.cuda.synchronize(); return time.perf_counter() - t0, r


only = sys.argv[1:] or ["pso", "cpso"]
for method in only:
    wall(method, 100)
    per = []
    for _ in range(3):
        t1, r1 = wall(method, 100); t2, r2 = wall(method, 1100)
        per.append((t2 - t1) / (r2.nit - r1.nit))
    print(f"C3 {method:4s} ackley n256 P16384: {min(per)*1e6:6.1f} us/gen -> {16384/min(per):.3e} evals/s   "
          f"(three measurements: {', '.join('%.1f' % (p*1e6) for p in per)} us; fun {r2.fun:.6g})", flush=True)
