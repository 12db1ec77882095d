"""Where a device-resident CMA-ES run and the oracle (LAPACK + canonical signs) part ways, and what the oracle's model looked
like there: per generation the deviation (best-f, best-x), the margin of the hsig test (cmaes/_cmaes.py:283-285: a branch), the
smallest relative gap between neighbouring eigenvalues of C, and the smallest gap between neighbouring fitness values around the
selection boundary mu (a swap there changes which candidate enters the mean).  usage: c4_divergence.py [gens seed n P objective]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
import stochopy_amd as sa

gens = int(sys.argv[1]) if len(sys.argv) > 1 else 16
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n, P = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (512, 1024)
objective = sys.argv[5] if len(sys.argv) > 5 else "rosenbrock"
opts = {"maxiter": gens, "popsize": P, "seed": seed, "sigma": 0.1, "ftol": -1.0, "xtol": 0.0, "return_all": True, "verbosity": 0.0}
bounds = [[-5.12, 5.12]] * n
from oracle import engine as oe
k = oe.cma_constants(n, P, 0.5)
rows = {}


def probe(it, before, after):
    ps = after["ps"]
    hs = np.linalg.norm(ps) / np.sqrt(1.0 - (1.0 - k["cs"]) ** (2.0 * it)) / k["chind"] - (1.4 + 2.0 / (n + 1.0))
    lam = after["D"] ** 2
    gap = np.diff(np.sort(lam)).min() / lam.max()
    f = np.sort(after["arfit"])
    mu = k["mu"]
    fgap = np.abs(np.diff(f)).min() / abs(f[0])
    rows[it] = (hs, gap, fgap, after["sigma"])


ref = oracle.minimize(objective, bounds, method="cmaes", options=dict(opts, eigh="canonical", probe=probe), rng="philox")
got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
rel = np.abs(got.funall[:, 0] / ref.funall[:, 0] - 1.0)
dx = np.abs(got.xall[:, 0, :] - ref.xall[:, 0, :]).max(axis=1)
print(f"{objective} n={n} P={P} seed {seed}: generation, |best-f/ref-1|, max|best-x-ref|, oracle: hsig margin, min eigen gap / max, min fitness gap / f0, sigma")
for g in range(gens):
    hs, gap, fgap, sg = rows[g + 1]
    print("  %3d  %.2e  %.2e   hsig %+.3e   eig gap %.2e   fit gap %.2e   sigma %.6g" % (g + 1, rel[g], dx[g], hs, gap, fgap, sg))
