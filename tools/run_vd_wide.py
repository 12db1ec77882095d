"""VD-CMA at a wide dimension for profiling: python tools/run_vd_wide.py [n] [P] [maxiter]"""
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa
n, P, it = (int(a) for a in (sys.argv[1:] + ["16384", "1024", "40"])[:3])
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, sigma=0.3, maxiter=it)
r = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma", options=o)
print(r.nit, r.fun)
