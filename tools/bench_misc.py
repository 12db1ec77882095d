"""Throughput of the claimed-but-unprofiled paths: BASELINE config 2 (DE Rastrigin n=128 P=4096), VD-CMA at n=4096,
CMA-ES with constraints="Penalize".  Per-generation cost from two run lengths; run under rocprofv3 for kernel stats."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


def timed(label, make, short, long_):
    wall(lambda: make(short))
    t1, r1 = wall(lambda: make(short)); t2, r2 = wall(lambda: make(long_))
    per = (t2 - t1) / (r2.nit - r1.nit)
    print(f"{label}: {per*1e6:.1f} us/gen -> {r2.nfev/r2.nit/per:.3e} evals/s   ({r1.nit} gens {t1*1e3:.1f} ms, {r2.nit} gens {t2*1e3:.1f} ms; fun {r2.fun:.6g})", flush=True)


b = lambda n: [[-5.12, 5.12]] * n
o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}
timed("C2 de rastrigin n128 P4096", lambda m: sa.optimize.minimize(sa.factory.rastrigin, b(128), method="de", options=dict(o, maxiter=m)), 200, 2200)
timed("M  de rosenbrock n128 P4096", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b(128), method="de", options=dict(o, maxiter=m)), 200, 2200)
ov = {"seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0}
for n, P in ((4096, 64), (4096, 1024)):
    timed(f"vdcma rosenbrock n{n} P{P}", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b(n), method="vdcma", options=dict(ov, popsize=P, maxiter=m)), 10, 60)
timed("cmaes Penalize sphere n128 P256 (bounds [1,5]: the mean sits outside the box)", lambda m: sa.optimize.minimize(
    sa.factory.sphere, [[1.0, 5.0]] * 128, method="cmaes", options=dict(ov, popsize=256, maxiter=m, constraints="Penalize", sigma=0.3)), 10, 60)
