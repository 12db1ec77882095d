"""Round 5, SURVEY.md section 8(f) rows on the current binary: whole-generation lines for VD-CMA (narrow
and wide rows), updating="immediate" and NA.  Run plain for the lines, under rocprofv3 (tools/prof_cmd.sh) for kernel stats.
Usage: python tools/r5_frows.py [vdcma] [immediate] [na]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import torch

import stochopy_amd as sa

which = set(sys.argv[1:]) or {"vdcma", "immediate", "na"}
b = lambda n: [[-5.12, 5.12]] * n  # noqa: E731


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


def per_gen(make, short, long_):
    wall(lambda: make(short))
    (t1, r1), (t2, r2) = wall(lambda: make(short)), wall(lambda: make(long_))
    return (t2 - t1) / (r2.nit - r1.nit), r2


if "vdcma" in which:
    ov = {"seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "backend": "hip", "sigma": 0.3}
    for n, P in ((512, 1024), (4096, 1024), (16384, 1024), (16384, 2048), (16384, 4096), (65536, 256)):
        t, r = per_gen(lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b(n), method="vdcma", options=dict(ov, popsize=P, maxiter=m)),
                       10, 50)
        print(f"vdcma rosenbrock n={n:6d} P={P:5d}: {t*1e6:9.1f} us/generation  {P/t:.3e} evals/s  "
              f"{32*n*P/t/1e9:8.1f} GB/s = {32*n*P/t/8e12:.3f} of 8 TB/s on 32 n B per candidate", flush=True)

if "immediate" in which:
    for method, name, n, P in (("de", "rosenbrock", 128, 4096), ("pso", "ackley", 256, 16384), ("cpso", "ackley", 256, 16384),
                               ("de", "rosenbrock", 1024, 4096)):
        o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "backend": "hip", "strict_updating": True}
        t, r = per_gen(lambda m: sa.optimize.minimize(getattr(sa.factory, name), b(n), method=method, options=dict(o, maxiter=m)), 3, 11)
        print(f"immediate {method:4s} {name:10s} n={n:5d} P={P:6d}: {t*1e3:9.3f} ms/generation = {t/P*1e6:7.3f} us/individual, "
              f"{P/t:10.3e} evals/s", flush=True)

if "na" in which:
    for name, n, P, nr in (("rosenbrock", 16, 1024, 0.5), ("rosenbrock", 64, 4096, 0.5), ("ackley", 32, 2048, 0.25)):
        o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "backend": "hip", "nrperc": nr}
        t, r = per_gen(lambda m: sa.optimize.minimize(getattr(sa.factory, name), b(n), method="na", options=dict(o, maxiter=m)), 5, 25)
        print(f"na {name:10s} n={n:4d} P={P:5d} nrperc={nr}: {t*1e3:9.3f} ms/generation  {P/t:.3e} evals/s", flush=True)
