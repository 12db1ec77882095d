"""CPSO at BASELINE config 3b (Ackley n=256, P=16384, Philox): cost per generation over a long run (the restart test
runs every generation, fires in some) and while the restart fires every generation (short maxiter), PSO next to it.
Wall clock around whole minimize() calls, two run lengths, minimum of three."""
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


def per_gen(method, short, long):
    best = None
    for _ in range(3):
        t1, r1 = wall(method, short); t2, r2 = wall(method, long)
        v = (t2 - t1) / (r2.nit - r1.nit) * 1e6
        best = v if best is None else min(best, v)
    return best


wall("cpso", 20)
for _ in range(2):
    v = per_gen("cpso", 200, 1200)
    print(f"cpso: long run {v:6.1f} us/gen ({16384 / v * 1e6:.3e} evals/s)   restart firing every generation "
          f"{per_gen('cpso', 20, 120):6.1f} us/gen", flush=True)
print(f"pso (no restart test) for comparison: {per_gen('pso', 200, 1200):6.1f} us/gen")
