"""One short wide VD-CMA run for rocprofv3 --kernel-trace: n P gens from the command line."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stochopy_amd as sa  # noqa: E402

n, P, gens = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (16384, 1024, 40)
opts = {"maxiter": gens, "popsize": P, "seed": 3, "sigma": 0.3, "backend": "hip", "rng": "philox", "xtol": 0.0, "ftol": -1.0}
res = sa.optimize.minimize(sa.factory.rosenbrock, [[-3.0, 3.0]] * n, method="vdcma", options=opts)
print(res.nit, res.fun)
