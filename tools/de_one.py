"""One DE (or PSO) shape for rocprofv3 --pmc passes: python tools/de_one.py method fun n P [generations]"""
import sys
sys.path.insert(0, "/root/repo")
import stochopy_amd as sa

method, name, n, P = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
gens = int(sys.argv[5]) if len(sys.argv) > 5 else 60
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, updating="deferred", maxiter=gens)
if method == "de":
    o["strategy"] = "best1bin"
r = sa.optimize.minimize(getattr(sa.factory, name), [[-5.12, 5.12]] * n, method=method, options=o)
print("done", method, name, n, P, r.nit)
