"""Randomised parity for the PSO / CPSO graph path (replayed hipGraphs: whole-batch kernels, generation-side swarm
radius, restarts re-seeded by the next generation kernel) against the oracle in Philox mode, bit for bit, and against
the generation-by-generation path (history).  usage: fuzz_cpso_graph.py [seconds] [seed]"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import oracle
import stochopy_amd as sa

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
bad = cases = restarts = 0
t0 = time.time()
while time.time() - t0 < budget:
    method = str(rs.choice(["cpso", "cpso", "cpso", "pso"]))
    n = int(rs.choice([1, 2, 7, 16, 33, 64, 64, 100, 128, 128, 129, 200, 256, 256, 257, 300, 520]))
    P = int(rs.randint(6, 300)) if rs.rand() < 0.8 else int(rs.randint(300, 2500))
    gens = int(rs.randint(16, 140))
    objective = str(rs.choice(["sphere", "rosenbrock"])) if n > 1 else "sphere"
    o = {"popsize": P, "maxiter": gens, "seed": int(rs.randint(1 << 30)), "updating": "deferred",
         "inertia": float(rs.uniform(0.4, 0.95))}
    if rs.rand() < 0.25:
        o.update(ftol=float(10 ** rs.uniform(-2, 3)), xtol=float(10 ** rs.uniform(-3, 1)))
    else:
        o.update(ftol=-1.0, xtol=0.0)
    if rs.rand() < 0.5:
        o["constraints"] = "Shrink"
    if method == "cpso":
        o["competitivity"] = float(rs.uniform(0.4, 1.6))
    lo, hi = (-5.12, 5.12) if rs.rand() < 0.7 else (-0.5, 0.8)
    b = [[lo, hi]] * n
    ref = oracle.minimize(objective, b, method=method, options=dict(o), rng="philox")
    restarts += len(ref.get("_restarts", []))
    got = sa.optimize.minimize(getattr(sa.factory, objective), b, method=method, options=dict(o, backend="hip", rng="philox"))
    hist = sa.optimize.minimize(getattr(sa.factory, objective), b, method=method,
                                options=dict(o, backend="hip", rng="philox", return_all=True, verbosity=0.0))
    ok = all((r.nit, r.nfev, r.status) == (ref["nit"], ref["nfev"], ref["status"]) and np.array_equal(r.x, ref["x"])
             and r.fun == ref["fun"] for r in (got, hist))
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH", method, objective, n, P, gens, o, got.nit, hist.nit, ref["nit"], got.fun, hist.fun, ref["fun"], flush=True)
print(f"{cases} cases ({restarts} restarts in the oracle runs), {bad} mismatches, {time.time() - t0:.0f} s")
