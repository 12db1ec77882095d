import sys; sys.path.insert(0, "/root/repo")
import numpy as np
import stochopy_amd as sa, oracle
n, P = 128, 4096
bounds = [[-5.12, 5.12]] * n
for maxiter in (3, 12, 51, 130):
    o = {"maxiter": maxiter, "popsize": P, "seed": 5, "updating": "deferred", "backend": "hip", "rng": "philox", "ftol": -1.0, "xtol": 0.0}
    a = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o))
    b = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o), callback=lambda X, r: None)
    print(maxiter, a.nit, b.nit, a.fun, b.fun, a.fun == b.fun, np.array_equal(a.x, b.x), a.status, b.status)
