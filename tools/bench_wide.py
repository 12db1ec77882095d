"""Wide rows (n > 4096, csrc/sx_wide.hip): sx_eval, DE, PSO and VD-CMA generations -- us per launch / generation and the
fraction of the HBM peak on the ALGORITHMIC bytes (eval 8n + 8 B per evaluation; DE (k + 2) rows + 16 B;
PSO X, V, pbest read + X, V written + 16 B = 40 n + 16 (pbest rewritten only when it improves); VD-CMA: see DESIGN.md).
Usage: python tools/bench_wide.py [eval] [de] [pso] [vdcma] [de16] [pso16]"""
import sys
import time

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import stochopy_amd as sa
from stochopy_amd import _device, _lib

which = set(sys.argv[1:]) or {"eval", "de", "pso", "vdcma"}
PEAK = 8000.0


def per_gen(method, fun, n, opts, short, long_, reps=2):
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


if "eval" in which:
    ctx = _device.Context()
    for name, n, P in (("rosenbrock", 4097, 1 << 15), ("rosenbrock", 8192, 1 << 14), ("rosenbrock", 16384, 1 << 13),
                       ("rosenbrock", 65536, 1 << 11), ("sphere", 16384, 1 << 13), ("ackley", 16384, 1 << 13),
                       ("rastrigin", 8192, 1 << 14), ("rosenbrock", 4096, 1 << 15), ("rosenbrock", 2048, 1 << 16),
                       ("rosenbrock", 16384, 64)):
        X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
        f = ctx.empty((P,))
        fid = _lib.FUN_IDS[name]
        with torch.cuda.stream(ctx.stream):
            for _ in range(3):
                _device.evaluate(ctx, fid, X, n, f=f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 20
            e0.record(ctx.stream)
            for _ in range(reps):
                _device.evaluate(ctx, fid, X, n, f=f)
            e1.record(ctx.stream)
            ctx.sync()
        us = e0.elapsed_time(e1) / reps * 1e3
        byts = (8 * n + 8) * P
        print(f"sx_eval {name:11s} n={n:6d} P={P:7d}: {us:9.1f} us  {P/us*1e6:.3e} evals/s  {byts/us/1e3:8.1f} GB/s "
              f"({byts/us/1e3/PEAK:.2f} of 8 TB/s)", flush=True)
        del X

if "de" in which:
    for name, n, P, strat in (("rosenbrock", 8192, 4096, "best1bin"), ("rosenbrock", 16384, 2048, "best1bin"),
                              ("rosenbrock", 32768, 1024, "best1bin"), ("rosenbrock", 65536, 1024, "best1bin"),
                              ("rastrigin", 8192, 4096, "rand1bin"), ("rosenbrock", 4096, 8192, "best1bin")):
        t = per_gen("de", getattr(sa.factory, name), n, {"popsize": P, "updating": "deferred", "strategy": strat}, 20, 120)
        k = _lib.DE_DONORS[strat]
        byts = (8 * n * (k + 2) + 16) * P
        print(f"DE {strat} {name:10s} n={n:6d} P={P:6d}: {t*1e6:9.1f} us/generation  {P/t:.3e} evals/s  "
              f"{byts/t/1e9:8.1f} GB/s ({byts/t/1e9/PEAK:.2f} of 8 TB/s)", flush=True)

if "degen" in which:  # strategies / constraints without a kernel of their own
    for name, n, P, strat, cons in (("rosenbrock", 8192, 4096, "rand2bin", None), ("rosenbrock", 8192, 4096, "best1bin", "Random"),
                                    ("rastrigin", 8192, 4096, "best2bin", None), ("rosenbrock", 32768, 1024, "rand2bin", None)):
        o = {"popsize": P, "updating": "deferred", "strategy": strat}
        if cons:
            o["constraints"] = cons
        t = per_gen("de", getattr(sa.factory, name), n, o, 20, 120)
        k = _lib.DE_DONORS[strat]
        byts = (8 * n * (k + 2) + 16) * P
        print(f"DE {strat} {cons} {name:10s} n={n:6d} P={P:6d}: {t*1e6:9.1f} us/generation  {byts/t/1e9/PEAK:.2f} of 8 TB/s", flush=True)
if "de16" in which:  # rows whose LDS leaves one workgroup per CU
    for name, n, P, strat in (("rosenbrock", 12000, 3072, "best1bin"), ("rosenbrock", 16384, 2048, "best1bin"),
                              ("rastrigin", 16384, 2048, "rand1bin"), ("sphere", 18000, 2048, "best1bin")):
        t = per_gen("de", getattr(sa.factory, name), n, {"popsize": P, "updating": "deferred", "strategy": strat}, 20, 120, reps=3)
        k = _lib.DE_DONORS[strat]
        byts = (8 * n * (k + 2) + 16) * P
        print(f"DE {strat} {name:10s} n={n:6d} P={P:6d}: {t*1e6:9.1f} us/generation  {byts/t/1e9/PEAK:.2f} of 8 TB/s", flush=True)
if "pso16" in which:
    for name, n, P in (("rosenbrock", 12000, 3072), ("rosenbrock", 16384, 2048), ("ackley", 16384, 2048)):
        t = per_gen("pso", getattr(sa.factory, name), n, {"popsize": P, "updating": "deferred"}, 20, 120, reps=3)
        byts = (40 * n + 16) * P
        print(f"PSO {name:10s} n={n:6d} P={P:6d}: {t*1e6:9.1f} us/generation  {byts/t/1e9/PEAK:.2f} of 8 TB/s", flush=True)

if "pso" in which:
    for name, n, P, extra in (("ackley", 8192, 4096, {}), ("rosenbrock", 16384, 2048, {}), ("rosenbrock", 65536, 512, {}),
                              ("rosenbrock", 8192, 4096, {"constraints": "Shrink"}), ("ackley", 4096, 8192, {})):
        t = per_gen("pso", getattr(sa.factory, name), n, dict({"popsize": P, "updating": "deferred"}, **extra), 20, 120)
        byts = (40 * n + 16) * P
        print(f"PSO {name:10s} n={n:6d} P={P:6d} {extra}: {t*1e6:9.1f} us/generation  {P/t:.3e} evals/s  "
              f"{byts/t/1e9:8.1f} GB/s ({byts/t/1e9/PEAK:.2f} of 8 TB/s)", flush=True)

if "vdcma" in which:
    for n, P in ((16384, 1024), (16384, 4096), (65536, 512), (16384, 33), (4096, 4096)):
        t = per_gen("vdcma", sa.factory.rosenbrock, n, {"popsize": P, "sigma": 0.3}, 10, 60)
        # algorithmic bytes of a generation: y and x of every candidate written (16 n), x read by the objective
        # (8 n), x and y of the mu = P/2 selected rows read by the moment sums (8 n per candidate on average)
        byts = 32 * n * P
        print(f"VD-CMA rosenbrock n={n:6d} P={P:6d}: {t*1e6:9.1f} us/generation  {P/t:.3e} evals/s  "
              f"{byts/t/1e9:8.1f} GB/s ({byts/t/1e9/PEAK:.2f} of 8 TB/s on 32 n B per candidate)", flush=True)
