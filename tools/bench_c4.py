"""BASELINE config 4 (CMA-ES Rosenbrock n=512 P=1024, Philox draws): per-generation cost of the device-resident loop
(default with rng="philox") and of the host-driven loop (callback given), from two run lengths.  usage: bench_c4.py [short long]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa

short, long_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 60)


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


def timed(label, make):
    wall(lambda: make(short))
    t1, r1 = wall(lambda: make(short)); t2, r2 = wall(lambda: make(long_))
    per = (t2 - t1) / (r2.nit - r1.nit)
    print(f"{label}: {per*1e3:.3f} ms/gen -> {r2.nfev/r2.nit/per:.3e} evals/s   ({r1.nit} gens {t1*1e3:.1f} ms, {r2.nit} gens {t2*1e3:.1f} ms; fun {r2.fun:.6g})", flush=True)


b512 = [[-5.12, 5.12]] * 512
o = {"popsize": 1024, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0}
timed("C4 cmaes rosenbrock n512 P1024, device-resident loop", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b512, method="cmaes", options=dict(o, maxiter=m)))
timed("C4 same with a callback (device-resident loop, every generation shown to the host)", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b512, method="cmaes", options=dict(o, maxiter=m), callback=lambda X, r: None))
os.environ["SX_CMA_LOOP"] = "host"
timed("C4 same, host-driven loop (SX_CMA_LOOP=host, callback)", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b512, method="cmaes", options=dict(o, maxiter=m), callback=lambda X, r: None))
del os.environ["SX_CMA_LOOP"]
for n, P in ((128, 256), (256, 512), (1024, 2048)):
    b = [[-5.12, 5.12]] * n
    oo = dict(o, popsize=P)
    timed(f"cmaes rosenbrock n{n} P{P}, device-resident loop", lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b, method="cmaes", options=dict(oo, maxiter=m)))
