"""Round 4: per-generation time of the row kernels for the library given through tools/ab_lib.py (streaming stores / loads
A/B): M, C2, the C5 shard shape (DE n=1024 P=16384), C3a (PSO) and C3b (CPSO), Philox draws, two run lengths, best of three.
usage: python tools/ab_lib.py <lib.so> tools/nt_ab.py [de|c5|pso ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa


def per_gen(method, obj, n, P, short, long_, **extra):
    b = [[-5.12, 5.12]] * n
    o = dict({"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}, **extra)

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(getattr(sa.factory, obj), b, method=method, options=dict(o, maxiter=m))
        torch.cuda.synchronize(); return time.perf_counter() - t0, r

    wall(short)
    per = []
    for _ in range(3):
        (t1, r1), (t2, r2) = wall(short), wall(long_)
        per.append((t2 - t1) / (r2.nit - r1.nit))
    t = min(per)
    print(f"{os.path.basename(sa._lib.LIB_PATH):22s} {method:5s} {obj:10s} n{n} P{P}: {t*1e6:7.3f} us/gen -> {P/t:.3e} evals/s (fun {r2.fun!r})", flush=True)


which = sys.argv[1:] or ["de", "c5", "pso"]
if "de" in which:
    per_gen("de", "rosenbrock", 128, 4096, 1000, 9000, strategy="best1bin")
    per_gen("de", "rastrigin", 128, 4096, 1000, 9000, strategy="best1bin")
if "c5" in which:
    per_gen("de", "rosenbrock", 1024, 16384, 100, 600, strategy="best1bin")
    per_gen("de", "rosenbrock", 1024, 16384, 100, 600, strategy="rand1bin")
if "pso" in which:
    per_gen("pso", "ackley", 256, 16384, 100, 1100)
    per_gen("cpso", "ackley", 256, 16384, 100, 1100)
