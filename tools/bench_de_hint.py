"""Metric shape (DE best1bin Rosenbrock n=128 P=4096, Philox): the chained kernel launched with the generation number
(sx_de_chain_run: SX_DE_HINT=1) against replayed graphs of the same kernel without it (SX_DE_HINT=0), per generation
from two run lengths, best of three, alternating.  Also C2 (Rastrigin).  usage: bench_de_hint.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa


def per_gen(obj, hint, short=1000, long_=9000):
    os.environ["SX_DE_HINT"] = "1" if hint else "0"
    b = [[-5.12, 5.12]] * 128
    o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred", "strategy": "best1bin"}

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(getattr(sa.factory, obj), b, method="de", options=dict(o, maxiter=m))
        torch.cuda.synchronize(); return time.perf_counter() - t0, r

    wall(short)
    best = None
    for _ in range(3):
        (t1, r1), (t2, r2) = wall(short), wall(long_)
        v = (t2 - t1) / (r2.nit - r1.nit)
        best = v if best is None or v < best else best
    return best, r2.fun


for obj in ("rosenbrock", "rastrigin"):
    for hint in (False, True, False, True):
        t, f = per_gen(obj, hint)
        print(f"DE {obj:10s} n128 P4096 {'hinted launches ' if hint else 'replayed graphs '}: {t*1e6:6.3f} us/gen -> {4096/t:.3e} evals/s (fun {f:.6g})", flush=True)
