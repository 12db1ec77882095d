"""M (DE best1bin Rosenbrock n=128 P=4096, Philox): cost per generation against the length of the replayed hipGraph
(_DeRun.GRAPH_CHUNK), from two run lengths.  usage: de_chunk_probe.py [chunk ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa
from stochopy_amd.optimize import _de

b = [[-5.12, 5.12]] * 128
o = {"popsize": 4096, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "updating": "deferred", "strategy": "best1bin"}


def wall(m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(o, maxiter=m))
    torch.cuda.synchronize()
    return time.perf_counter() - t0, r


for chunk in [int(a) for a in sys.argv[1:]] or [50, 100, 200, 400, 50, 200]:
    _de._DeRun.GRAPH_CHUNK = chunk
    short, long_ = 2000, 12000
    wall(short)
    best = 1e9
    for _ in range(3):
        t1, r1 = wall(short); t2, r2 = wall(long_)
        best = min(best, (t2 - t1) / (r2.nit - r1.nit))
    print("GRAPH_CHUNK %4d: %.3f us per generation -> %.4e evals/s" % (chunk, best * 1e6, 4096 / best), flush=True)
