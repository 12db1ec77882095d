"""Host time of one sx_vdcma_generation call vs device time (wide models)."""
import sys, time, ctypes as C
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from stochopy_amd import _lib, _device
from stochopy_amd.optimize import _vdcma
n, P = (int(a) for a in (sys.argv[1:] + ["16384", "33"])[:2])
lower, upper = np.full(n, -5.12), np.full(n, 5.12)
run = _vdcma._VdDeviceRun(_lib.FUN_IDS["rosenbrock"], lower, upper, None, 10000, P, 0.3, 0.5, 0.0, -1.0, 0, run=False)
ctx = run.ctx
with torch.cuda.stream(ctx.stream):
    for g in range(1, 6):
        run.step(g)
    ctx.sync()
    t0 = time.perf_counter()
    for g in range(6, 56):
        run.step(g)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
print(f"n={n} P={P}: host enqueue {1e6*(t1-t0)/50:.1f} us/generation, until drained {1e6*(t2-t0)/50:.1f} us/generation")
