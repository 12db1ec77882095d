"""One VD-CMA shape for rocprofv3 passes: python tools/vd_one.py n P [generations]"""
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa

n, P = int(sys.argv[1]), int(sys.argv[2])
gens = int(sys.argv[3]) if len(sys.argv) > 3 else 40
r = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma",
                         options=dict(seed=0, rng="philox", backend="hip", popsize=P, sigma=0.3, maxiter=gens))
print("done", n, P, r.nit)
