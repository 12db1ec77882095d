"""Round 6: the eigensolver's two ways of enqueueing a run -- one launch per round (sx_eigh_set_flow(0)) and one resident launch
(1) -- on the covariance matrices of a C4-like CMA-ES run (warm-started), and BASELINE config 4 itself under both.
usage: bench_eigh_flow.py [n P gens]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import oracle
import stochopy_amd as sa
from stochopy_amd import _device, _lib
from stochopy_amd.linalg import Eigh

ctx = _device.Context()
L = _lib.lib()
n, P, gens = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 1024, 8)
rec = []
def record(C):
    w, V = np.linalg.eigh(C); rec.append((C.copy(), V)); return w, V
oracle.minimize("rosenbrock", [[-5.12, 5.12]] * n, method="cmaes",
                options={"popsize": P, "seed": 0, "maxiter": gens, "ftol": -1.0, "xtol": 0.0, "eigh": record}, rng="philox")
tol = max(1e-14, n * 1.1102230246251565e-16)
eig = Eigh(ctx, n)
for g in range(1, len(rec)):
    Cd, Sd = ctx.upload(rec[g][0]), ctx.upload(np.ascontiguousarray(rec[g - 1][1]))
    res = {}
    for mode in (0, 1):
        L.sx_eigh_set_flow(mode)
        ts = []
        for rep in range(4):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            w, B = eig(Cd, tol=tol, start=Sd, refine=True)
            ctx.sync(); ts.append(time.perf_counter() - t0)
        sw, conv, off = eig.info()
        res[mode] = (min(ts[1:]), sw, conv, off, w.cpu().numpy().copy(), B.cpu().numpy().copy())
    same = np.array_equal(res[0][4], res[1][4]) and np.array_equal(res[0][5], res[1][5])
    print("generation %2d: per round %.3f ms (sweeps %d conv %d) | resident %.3f ms (sweeps %d conv %d) | identical %s"
          % (g + 1, res[0][0] * 1e3, res[0][1], res[0][2], res[1][0] * 1e3, res[1][1], res[1][2], same), flush=True)
L.sx_eigh_set_flow(-1)


def wall(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0, r


b = [[-5.12, 5.12]] * n
o = {"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0}
for mode in (0, 1, 0, 1):
    L.sx_eigh_set_flow(mode)
    mk = lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b, method="cmaes", options=dict(o, maxiter=m))
    wall(lambda: mk(10))
    t1, r1 = wall(lambda: mk(10)); t2, r2 = wall(lambda: mk(60))
    per = (t2 - t1) / (r2.nit - r1.nit)
    print("C4 cmaes rosenbrock n%d P%d, flow=%d: %.3f ms/gen -> %.3e evals/s (10 gens %.1f ms, 60 gens %.1f ms; fun %.10g)"
          % (n, P, mode, per * 1e3, r2.nfev / r2.nit / per, t1 * 1e3, t2 * 1e3, r2.fun), flush=True)
L.sx_eigh_set_flow(-1)
