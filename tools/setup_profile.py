"""Where the fixed cost of a minimize() call goes (DE at the metric shape, one generation)."""
import sys, time, cProfile, pstats
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
b = [[-5.12, 5.12]] * 128
o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred", "strategy": "best1bin"}
run = lambda m: sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(o, maxiter=m))
run(60); run(1)
pr = cProfile.Profile(); pr.enable(); run(1); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
