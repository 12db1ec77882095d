import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa
r = sa.optimize.minimize(sa.factory.ackley, [[-5.12, 5.12]] * 256, method="cpso",
                         options={"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": 80, "updating": "deferred"})
print(r.nit, r.fun)
