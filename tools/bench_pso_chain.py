"""BASELINE config 3a (PSO Ackley n=256 P=16384): the chained kernel (one kernel per generation) against the
generation + select_finalize pair, per generation from two run lengths, best of three.  usage: bench_pso_chain.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa


def per_gen(chain, short=400, long_=2400):
    os.environ["SX_PSO_CHAIN"] = "1" if chain else "0"
    b = [[-5.12, 5.12]] * 256
    o = {"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(sa.factory.ackley, b, method="pso", options=dict(o, maxiter=m))
        torch.cuda.synchronize(); return time.perf_counter() - t0, r

    wall(short)
    best = None
    for _ in range(3):
        (t1, r1), (t2, r2) = wall(short), wall(long_)
        v = (t2 - t1) / (r2.nit - r1.nit)
        best = v if best is None or v < best else best
    return best, r2.fun


for chain in (False, True, False, True):
    t, f = per_gen(chain)
    print(f"PSO C3a {'chained (1 kernel/gen)' if chain else 'two kernels/gen      '}: {t*1e6:6.2f} us/gen -> {16384/t:.3e} evals/s (fun {f:.6g})", flush=True)
