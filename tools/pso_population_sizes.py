"""PSO / CPSO generations over population sizes: us per generation and the fraction of the HBM peak on (48 n + 24) B per evaluation."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa


def per_gen(method, fun, n, P, short, long_, reps=2, **extra):
    o = dict(dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, updating="deferred"), **extra)

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(fun, [[-5.12, 5.12]] * n, method=method, options=dict(o, maxiter=m))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r.nit
    wall(short)
    runs = [(wall(short), wall(long_)) for _ in range(reps)]
    (t1, n1), (t2, n2) = min(r[0] for r in runs), min(r[1] for r in runs)
    return (t2 - t1) / (n2 - n1)


for method, name, n, P in (("pso", "ackley", 256, 16384), ("pso", "ackley", 256, 1 << 18), ("pso", "rosenbrock", 128, 16384),
                           ("pso", "rosenbrock", 128, 1 << 18), ("pso", "rosenbrock", 128, 1 << 20), ("pso", "rosenbrock", 64, 1 << 20),
                           ("pso", "rosenbrock", 100, 1 << 18), ("pso", "rosenbrock", 200, 1 << 18), ("pso", "rosenbrock", 300, 1 << 17),
                           ("pso", "rosenbrock", 1024, 1 << 16), ("cpso", "ackley", 256, 1 << 18), ("pso", "rosenbrock", 256, 1 << 18)):
    gens = max(40, min(1500, int(0.15 / (6 * 8 * n * P / 4e12 + 3e-6))))
    t = per_gen(method, getattr(sa.factory, name), n, P, max(10, gens // 6), gens)
    print(f"{method:4s} {name:10s} n={n:5d} P={P:8d}: {t*1e6:9.2f} us/generation  {(48*n+24)*P/t/1e9/8000:.3f} of 8 TB/s", flush=True)
