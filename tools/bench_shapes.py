"""Shapes OFF the benchmark grid (VERDICT r4 missing #4 / next #3): DE best1bin and PSO generations at row lengths and
population sizes that no compile-time specialisation names, each beside the nearest specialised shape -- us per generation
(two run lengths of whole minimize() calls, set-up cancels) and the fraction of the HBM peak on the ALGORITHMIC bytes
(DE (k + 2) rows + 16 B per evaluation, PSO 48 n + 24).
Usage: python tools/bench_shapes.py [de] [pso] [large]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import torch

import stochopy_amd as sa
from stochopy_amd import _lib

which = set(sys.argv[1:]) or {"de", "pso", "large"}
PEAK = 8000.0


def per_gen(method, fun, n, opts, short, long_, reps=3):
    bounds = [[-5.12, 5.12]] * n
    o = dict(dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip"), **opts)

    def wall(m):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = sa.optimize.minimize(fun, bounds, method=method, options=dict(o, maxiter=m))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r.nit

    wall(short)
    runs = [(wall(short), wall(long_)) for _ in range(reps)]
    (t1, n1), (t2, n2) = min(r[0] for r in runs), min(r[1] for r in runs)
    return (t2 - t1) / (n2 - n1)


def lengths(n, P):
    # enough generations that the difference of the two runs is >= ~50 ms of device time
    per = max(2e-6, 3 * 8 * n * P / 5e12)
    long_ = int(min(4000, max(120, 0.08 / per)))
    return max(10, long_ // 6), long_


if "de" in which:
    # (off-grid shape, its specialised neighbour)
    pairs = [((100, 4096), (128, 4096)), ((130, 4096), (128, 4096)), ((128, 4000), (128, 4096)),
             ((300, 16384), (256, 16384)), ((1000, 16384), (1024, 16384)), ((1500, 8192), (2048, 8192)),
             ((64, 4096), (64, 4096)), ((200, 8192), (256, 8192)), ((700, 8192), (512, 8192))]
    done = {}
    for name in ("rosenbrock", "rastrigin"):
        for off, near in pairs:
            row = []
            for n, P in (off, near):
                key = (name, n, P)
                if key not in done:
                    s, l = lengths(n, P)
                    t = per_gen("de", getattr(sa.factory, name), n,
                                {"popsize": P, "updating": "deferred", "strategy": "best1bin"}, s, l)
                    done[key] = (t, (8 * n * 4 + 16) * P / t / 1e9 / PEAK)
                row.append(done[key])
            (t0, f0), (t1, f1) = row
            print(f"DE best1bin {name:10s} n={off[0]:5d} P={off[1]:6d}: {t0*1e6:8.2f} us/gen frac {f0:.3f}   | neighbour "
                  f"n={near[0]:5d} P={near[1]:6d}: {t1*1e6:8.2f} us/gen frac {f1:.3f}   ratio {f0/f1:.2f}", flush=True)

if "pso" in which:
    done = {}
    for name, off, near in (("ackley", (250, 16000), (256, 16384)), ("ackley", (100, 16384), (128, 16384)),
                            ("rosenbrock", (1000, 8192), (1024, 8192)), ("ackley", (300, 16384), (256, 16384))):
        row = []
        for n, P in (off, near):
            key = (name, n, P)
            if key not in done:
                s, l = lengths(n, P)
                t = per_gen("pso", getattr(sa.factory, name), n, {"popsize": P, "updating": "deferred"}, s, l)
                done[key] = (t, (48 * n + 24) * P / t / 1e9 / PEAK)
            row.append(done[key])
        (t0, f0), (t1, f1) = row
        print(f"PSO {name:10s} n={off[0]:5d} P={off[1]:6d}: {t0*1e6:8.2f} us/gen frac {f0:.3f}   | neighbour "
              f"n={near[0]:5d} P={near[1]:6d}: {t1*1e6:8.2f} us/gen frac {f1:.3f}   ratio {f0/f1:.2f}", flush=True)

if "large" in which:
    # the metric's row length without the launch-bound effects: n = 128, P = 2^20 (4.3 GB algorithmic per generation)
    for name, n, P in (("rosenbrock", 128, 1 << 20), ("rastrigin", 128, 1 << 20), ("rosenbrock", 128, 1 << 16),
                       ("rosenbrock", 64, 1 << 20), ("rosenbrock", 256, 1 << 19)):
        t = per_gen("de", getattr(sa.factory, name), n, {"popsize": P, "updating": "deferred", "strategy": "best1bin"},
                    10, 60, reps=2)
        byts = (8 * n * 4 + 16) * P
        print(f"DE best1bin {name:10s} n={n:5d} P={P:8d}: {t*1e6:9.1f} us/gen  {P/t:.3e} evals/s  "
              f"{byts/t/1e9:8.1f} GB/s ({byts/t/1e9/PEAK:.3f} of 8 TB/s)", flush=True)
