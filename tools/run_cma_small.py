"""One CMA-ES run at a small size for the profiler: python run_cma_small.py [n P maxiter]"""
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa
n, P, m = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (64, 128, 200)
r = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="cmaes",
                         options={"popsize": P, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": m})
print(r.nit, r.fun)
