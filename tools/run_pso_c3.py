"""One PSO run at BASELINE config 3a (Ackley n=256, P=16384, Philox) for the profiler: python run_pso_c3.py [maxiter]"""
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa
m = int(sys.argv[1]) if len(sys.argv) > 1 else 200
r = sa.optimize.minimize(sa.factory.ackley, [[-5.12, 5.12]] * 256, method="pso",
                         options={"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": m, "updating": "deferred"})
print(r.nit, r.fun)
