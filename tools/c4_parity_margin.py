"""How far the device-resident C4 run (CMA-ES Rosenbrock n=512 P=1024, Philox) is from the oracle (LAPACK + canonical signs),
generation by generation, with and without the eigensolver's refinement step: max |best-f / ref - 1| and max |x - x_ref|.
usage: c4_parity_margin.py [generations]  (run once per mode: SX_EIGH_REFINE=0 / 1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import stochopy_amd as sa

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n, P = 512, 1024
opts = {"maxiter": gens, "popsize": P, "seed": 0, "sigma": 0.1, "ftol": -1.0, "xtol": 0.0, "return_all": True, "verbosity": 0.0}
bounds = [[-5.12, 5.12]] * n
ref = oracle.minimize("rosenbrock", bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox")
got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
rel = np.abs(got.funall[:, 0] / ref.funall[:, 0] - 1.0)
dx = np.abs(got.xall[:, 0, :] - ref.xall[:, 0, :]).max(axis=1)
print("SX_EIGH_REFINE=%s  nit %d/%d" % (os.environ.get("SX_EIGH_REFINE", "(default: on in the loops)"), got.nit, ref.nit))
for g in list(range(0, gens, 5)) + [gens - 1]:
    print("  generation %3d: |best-f / ref - 1| = %.2e   max |best-x - ref| = %.2e" % (g + 1, rel[g], dx[g]))
print("  max over the run: %.2e  %.2e   (test tolerance: 1e-6, 1e-6 * 10.24)" % (rel.max(), dx.max()))
