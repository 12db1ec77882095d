"""A plain Python objective (SURVEY.md 8b case iii) that costs ~1 ms per call, through the serial host loop and through
host pools (options host_workers / host_backend = what the reference's joblib backends do, _common.py:38-43): seconds per
generation and speed-up over the serial loop, next to the workers' ideal.  Two objectives: one that holds the GIL for
its whole millisecond (pure-Python arithmetic: only processes help) and one that releases it (a sleep standing for an
external solver / numpy-heavy model: threads help as well).
Usage: python tools/bench_host_pool.py [popsize] [ndim]"""
import os
import sys
import time
import warnings

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import stochopy_amd as sa

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
cores = os.cpu_count() or 1


def gil_bound(x, loops):
    s = 0.0
    for _ in range(loops):  # ~1 ms of interpreter work
        s += 1e-9
    return float(np.sum(x * x)) + 0.0 * s


def gil_free(x, delay):
    time.sleep(delay)
    return float(np.sum(x * x))


# calibrate the interpreter loop to ~1 ms
t0 = time.perf_counter(); gil_bound(np.zeros(n), 20000); per = (time.perf_counter() - t0) / 20000
loops = max(1, int(1e-3 / per))


def per_generation(fun, args, gens, **pool):
    o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "backend": "hip", "updating": "deferred"}
    o.update(pool)

    def run(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            r = sa.optimize.minimize(fun, [[-5.12, 5.12]] * n, args=args, method="de", options=dict(o, maxiter=m))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r.nit

    run(2)
    (t1, n1), (t2, n2) = run(2), run(2 + gens)
    return (t2 - t1) / (n2 - n1)


print(f"host cores {cores}; DE best1bin n={n} P={P}; objective ~1 ms per call", flush=True)
for label, fun, args in (("GIL-bound (pure Python loop)", gil_bound, (loops,)), ("GIL-free (sleeps 1 ms)", gil_free, (1e-3,))):
    ts = per_generation(fun, args, 1)
    print(f"{label:30s} serial host loop            : {ts:8.3f} s/generation  ({ts / P * 1e3:.3f} ms per call)", flush=True)
    for backend, w in (("loky", min(cores, 16)), ("loky", min(cores, 64)), ("loky", cores), ("threading", min(cores, 16)),
                       ("threading", min(cores, 64))):
        if backend == "threading" and fun is gil_bound and w > 16:
            continue
        t = per_generation(fun, args, 3, host_workers=w, host_backend=backend)
        print(f"{label:30s} host_workers={w:4d} {backend:9s}: {t:8.3f} s/generation  speed-up {ts / t:6.1f} "
              f"= {ts / t / w:.2f} x workers", flush=True)
