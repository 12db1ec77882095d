"""BASELINE config 4 (CMA-ES Rosenbrock n=512 P=1024, Philox, device-resident loop): one run of `gens` generations (profilers)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stochopy_amd as sa
gens = int(sys.argv[1]) if len(sys.argv) > 1 else 40
o = {"popsize": 1024, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": gens}
r = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * 512, method="cmaes", options=o)
print(r.nit, r.fun)
