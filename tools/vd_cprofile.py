import cProfile, pstats, sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
n, P = 16384, 1024
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, sigma=0.3)
run = lambda m: sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma", options=dict(o, maxiter=m))
run(10); run(200)
torch.cuda.synchronize(); t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable(); r = run(200); torch.cuda.synchronize(); pr.disable()
print("wall", time.perf_counter() - t0, r.nit)
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
