"""How far the device-resident C4 run (CMA-ES Rosenbrock n=512 P=1024, Philox) is from the oracle (LAPACK + canonical signs),
generation by generation, with and without the eigensolver's refinement step: max |best-f / ref - 1| and max |x - x_ref|.
usage: c4_parity_margin.py [generations [seed [n [P [objective]]]]]  (run once per mode: SX_EIGH_REFINE=0 / 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import stochopy_amd as sa

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n, P = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 1024)
objective = sys.argv[5] if len(sys.argv) > 5 else "rosenbrock"
opts = {"maxiter": gens, "popsize": P, "seed": seed, "sigma": 0.1, "ftol": -1.0, "xtol": 0.0, "return_all": True, "verbosity": 0.0}
bounds = [[-5.12, 5.12]] * n
ref = oracle.minimize(objective, bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox")
got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
rel = np.abs(got.funall[:, 0] / ref.funall[:, 0] - 1.0)
dx = np.abs(got.xall[:, 0, :] - ref.xall[:, 0, :]).max(axis=1)
print("%s n=%d P=%d seed %d  SX_EIGH_REFINE=%s  nit %d/%d" % (objective, n, P, seed, os.environ.get("SX_EIGH_REFINE", "(default: on in the loops)"),
                                                          got.nit, ref.nit))
for g in (list(range(0, gens, 5)) + [gens - 1] if os.environ.get("C4_BRIEF") != "1" else []):
    print("  generation %3d: |best-f / ref - 1| = %.2e   max |best-x - ref| = %.2e" % (g + 1, rel[g], dx[g]))
print("  max over the run: %.2e  %.2e   (test tolerance: 1e-6, 1e-6 * 10.24)" % (rel.max(), dx.max()))
