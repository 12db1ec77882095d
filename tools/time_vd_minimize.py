import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
n, P = (int(a) for a in (sys.argv[1:] + ["16384", "1024"])[:2])
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, sigma=0.3)
for m in (10, 10, 60, 60, 200):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma", options=dict(o, maxiter=m))
    torch.cuda.synchronize(); print(m, r.nit, r.status, f"{(time.perf_counter()-t0)*1e3:.2f} ms")
