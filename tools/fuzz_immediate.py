"""Randomised parity: updating="immediate" sweeps (csrc/sx_async.hip) vs the oracle, Philox draws, bit for bit.
usage: fuzz_immediate.py [cases] [seed]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import oracle
import stochopy_amd as sa

import os
RNG = os.environ.get("RNG", "philox")  # or numpy-legacy: host draws replayed by csrc/sx_mt19937.cpp
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = 0
t0 = time.time()
for c in range(cases):
    method = rs.choice(["de", "de", "pso", "cpso"])
    n = int(rs.choice([1, 2, 3, 5, 8, 16, 17, 33, 64, 65, 100, 128, 129, 200, 300]))
    P = int(rs.randint(6, 420)) if rs.rand() < 0.9 else int(rs.randint(420, 2500))
    gens = int(rs.randint(2, 9))
    if RNG != "philox":
        P = min(P, 420)
    objective = str(rs.choice(["sphere", "rosenbrock"])) if n > 1 else "sphere"
    o = {"popsize": P, "maxiter": gens, "seed": int(rs.randint(1 << 30)), "updating": "immediate", "return_all": True}
    if rs.rand() < 0.3:   # let runs stop early now and then
        o.update(ftol=float(10 ** rs.uniform(-2, 3)), xtol=float(10 ** rs.uniform(-3, 1)))
    else:
        o.update(ftol=-1.0, xtol=0.0)
    if method == "de":
        o["strategy"] = str(rs.choice(["rand1bin", "rand2bin", "best1bin", "best2bin"]))
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
    got = sa.optimize.minimize(getattr(sa.factory, objective), b, method=method,
                               options=dict(o, backend="hip", rng=RNG, strict_updating=True))
    ok = ((got.nit, got.nfev, got.status) == (ref["nit"], ref["nfev"], ref["status"]) and np.array_equal(got.x, ref["x"])
          and got.fun == ref["fun"] and np.array_equal(got.xall, ref["xall"]) and np.array_equal(got.funall, ref["funall"]))
    if not ok:
        bad += 1
        print("MISMATCH", c, method, objective, n, P, gens, o, got.nit, ref["nit"], got.status, ref["status"], got.fun, ref["fun"], flush=True)
print(f"{cases} cases, {bad} mismatches, {time.time() - t0:.0f} s")
