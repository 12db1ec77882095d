"""Timing of the non-headline BASELINE configs (C3 PSO/CPSO Ackley n256 P16384, C4 CMA-ES Rosenbrock n512 P1024)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import stochopy_amd as sa

def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r

def timed(label, make, short, long):
    """per-generation cost from two run lengths (the host-side initial population is in both)."""
    wall(lambda: make(short))
    t1, r1 = wall(lambda: make(short)); t2, r2 = wall(lambda: make(long))
    per = (t2 - t1) / (r2.nit - r1.nit)
    print(f"{label}: {per*1e6:.1f} us/gen -> {r2.nfev/r2.nit/per:.3e} evals/s   (setup+{r1.nit} gens {t1*1e3:.1f} ms; fun {r2.fun:.6g})")

b256 = [[-5.12, 5.12]] * 256
for method in ("pso", "cpso"):
    o = {"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}
    timed(f"C3 {method} ackley n256 P16384", lambda m, method=method: sa.optimize.minimize(sa.factory.ackley, b256, method=method, options=dict(o, maxiter=m)), 100, 1100)
b512 = [[-5.12, 5.12]] * 512
o = {"popsize": 1024, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}
timed("C4 cmaes rosenbrock n512 P1024", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b512, method="cmaes", options=dict(o, maxiter=m)), 4, 14)
o = {"popsize": 1024, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "eigh": "device"}
timed("C4 cmaes rosenbrock n512 P1024 eigh=device", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b512, method="cmaes", options=dict(o, maxiter=m)), 4, 14)
