"""Randomised parity for the synchronous paths: fused kernels vs the oracle (Philox, bit for bit) and the
propose/select path around a caller-supplied objective vs the fused run.  usage: fuzz_deferred.py [cases] [seed]"""
import ctypes as C, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import torch
import oracle
import stochopy_amd as sa
from stochopy_amd import _lib

L = _lib.lib()


def device_fun(name):
    fid = _lib.FUN_IDS[name]

    def fun(X):
        P, n = X.shape
        f = torch.empty((P,), dtype=torch.float64, device=X.device)
        X = X.contiguous()
        L.sx_eval(fid, X.data_ptr(), P, n, n, None, None, f.data_ptr(), None, None,
                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return f
    return sa.factory.batched(fun)


import os
RNG = os.environ.get("RNG", "philox")  # or numpy-legacy: host draws replayed by csrc/sx_mt19937.cpp
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for c in range(cases):
    method = rs.choice(["de", "de", "pso", "cpso"])
    n = int(rs.choice([1, 2, 3, 5, 8, 16, 17, 33, 64, 65, 100, 128, 129, 200, 256, 257, 300, 513, 700]))
    P = int(rs.randint(6, 420)) if rs.rand() < 0.9 else int(rs.randint(420, 3000))
    gens = int(rs.randint(2, 40))
    if RNG != "philox":
        P, gens = min(P, 420), min(gens, 15)
    objective = str(rs.choice(["sphere", "rosenbrock"])) if n > 1 else "sphere"
    o = {"popsize": P, "maxiter": gens, "seed": int(rs.randint(1 << 30)), "updating": "deferred"}
    if rs.rand() < 0.3:
        o.update(ftol=float(10 ** rs.uniform(-2, 3)), xtol=float(10 ** rs.uniform(-3, 1)))
    else:
        o.update(ftol=-1.0, xtol=0.0)
    if rs.rand() < 0.5:
        o.update(return_all=True, verbosity=float(rs.choice([0.0, 0.4, 1.0])))
    if method == "de":
        o["strategy"] = str(rs.choice(["rand1bin", "rand2bin", "best1bin", "best2bin"]))
        if P - 1 < 5:
            continue
        o["mutation"] = float(rs.uniform(0.2, 1.6))
        o["recombination"] = float(rs.uniform(0.0, 1.0))
        if rs.rand() < 0.4:
            o["constraints"] = "Random"
    else:
        o["inertia"] = float(rs.uniform(0.4, 0.95))
        if rs.rand() < 0.5:
            o["constraints"] = "Shrink"
        if method == "cpso":
            o["competitivity"] = float(rs.uniform(0.5, 1.5))
    lo, hi = (-5.12, 5.12) if rs.rand() < 0.7 else (-0.5, 0.8)
    b = [[lo, hi]] * n
    ref = oracle.minimize(objective, b, method=method, options=dict(o), rng=RNG)
    got = sa.optimize.minimize(getattr(sa.factory, objective), b, method=method, options=dict(o, backend="hip", rng=RNG))
    ext = sa.optimize.minimize(device_fun(objective), b, method=method, options=dict(o, backend="hip", rng=RNG))
    ok = True
    for r in (got, ext):
        ok = ok and (r.nit, r.nfev, r.status) == (ref["nit"], ref["nfev"], ref["status"]) and np.array_equal(r.x, ref["x"]) \
            and r.fun == ref["fun"]
        if "xall" in ref:
            ok = ok and np.array_equal(r.xall, ref["xall"]) and np.array_equal(r.funall, ref["funall"])
    if not ok:
        bad += 1
        print("MISMATCH", c, method, objective, n, P, gens, o, got.nit, ext.nit, ref["nit"], got.fun, ext.fun, ref["fun"], flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
