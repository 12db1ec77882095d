"""Hunt for the device-loop Penalize mismatch the fuzzer saw (n = 2, small P): device-resident loop vs host-driven loop
(callback given) vs oracle, per generation."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, stochopy_amd as sa
warnings.simplefilter("ignore")
rs = np.random.RandomState(7)
found = 0
for trial in range(400):
    n = int(rs.randint(2, 4)); P = int(rs.randint(2 * n + 2, 4 * n + 8))
    obj = str(rs.choice(["rosenbrock", "sphere", "ackley"]))
    lo = float(rs.uniform(-2, 2)); b = [[lo, lo + float(rs.uniform(0.5, 4))]] * n
    o = {"maxiter": int(rs.randint(5, 30)), "popsize": P, "seed": int(rs.randint(1 << 30)), "sigma": float(rs.uniform(0.05, 0.5)),
         "constraints": "Penalize", "return_all": True, "verbosity": 1.0, "ftol": -1.0, "xtol": 0.0}
    ref = oracle.minimize(obj, b, method="cmaes", options=dict(o, eigh="canonical"), rng="philox")
    dev = sa.optimize.minimize(getattr(sa.factory, obj), b, method="cmaes", options=dict(o, backend="hip", rng="philox"))
    host = sa.optimize.minimize(getattr(sa.factory, obj), b, method="cmaes", options=dict(o, backend="hip", rng="philox", eigh="device"),
                                callback=lambda X, r: None)
    def first_bad(a, c):
        m = min(len(a), len(c))
        bad = [g for g in range(m) if not np.allclose(a[g], c[g], rtol=1e-6, atol=1e-300)]
        return bad[0] + 1 if bad else None
    fd, fh = first_bad(dev.funall, ref.funall), first_bad(host.funall, ref.funall)
    if fd or fh:
        found += 1
        print(f"trial {trial}: {obj} n={n} P={P} bounds {b[0]} {o}\n   first bad generation: device loop {fd}, host loop {fh}; nit dev/host/ref {dev.nit}/{host.nit}/{ref.nit}")
        g = (fd or fh) - 1
        print("   ref funall[g]", np.sort(ref.funall[g])[:6], "\n   dev funall[g]", np.sort(dev.funall[g])[:6], "\n   host funall[g]", np.sort(host.funall[g])[:6])
        if g > 0:
            print("   prev gen equal (dev)?", np.allclose(dev.funall[g-1], ref.funall[g-1], rtol=1e-9), "xall equal?", np.allclose(dev.xall[g-1], ref.xall[g-1], rtol=1e-9, atol=1e-12))
            print("   this gen xall equal (dev vs ref)?", np.allclose(dev.xall[g], ref.xall[g], rtol=1e-7, atol=1e-10))
        if found >= 4: break
print("trials done; mismatching cases:", found)
