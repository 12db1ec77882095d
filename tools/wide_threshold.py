"""Where do the one-workgroup-per-row kernels (csrc/sx_wide.hip) overtake the wavefront-per-row kernels?  Rows of 2049 ... 4096
elements through whichever family the loaded library dispatches them to (run once per build: tools/ab_lib.py <lib.so> this)."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
from stochopy_amd import _device, _lib

print("library:", _lib.LIB_PATH, " wide_from =", _lib.wide_from(), flush=True)
ctx = _device.Context()
PEAK = 8000.0


def per_gen(method, fun, n, opts, short, long_, reps=2):
    o = dict(dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip"), **opts)

    def wall(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = sa.optimize.minimize(fun, [[-5.12, 5.12]] * n, method=method, options=dict(o, maxiter=m))
        torch.cuda.synchronize()
        return time.perf_counter() - t0, r.nit
    wall(short)
    runs = [(wall(short), wall(long_)) for _ in range(reps)]
    (t1, n1), (t2, n2) = min(r[0] for r in runs), min(r[1] for r in runs)
    return (t2 - t1) / (n2 - n1)


NS = [int(a) for a in sys.argv[1:]] or [2049, 2304, 2560, 3072, 3584, 4096]
for n in NS:
    P = ((1 << 27) // n) // 64 * 64
    row = [f"n={n:5d}"]
    for name in ("rosenbrock", "ackley"):
        X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
        f = ctx.empty((P,))
        torch.cuda.synchronize()
        with torch.cuda.stream(ctx.stream):
            for _ in range(3):
                _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            for _ in range(20):
                _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
            e1.record(ctx.stream); ctx.sync()
        us = e0.elapsed_time(e1) / 20 * 1e3
        row.append(f"eval {name[:5]} {(8*n+8)*P/us/1e3/PEAK:.2f}")
        del X
    t = per_gen("de", sa.factory.rosenbrock, n, {"popsize": 8192, "updating": "deferred", "strategy": "best1bin"}, 10, 60)
    row.append(f"DE P=8192 {t*1e6:7.1f} us {(32*n+16)*8192/t/1e9/PEAK:.2f}")
    t = per_gen("pso", sa.factory.ackley, n, {"popsize": 8192, "updating": "deferred"}, 10, 60)
    row.append(f"PSO P=8192 {t*1e6:7.1f} us {(48*n+24)*8192/t/1e9/PEAK:.2f}")
    print("   ".join(row), flush=True)
