"""does the throughput mode's 32-bit uniforms (PSO r1, r2; DE crossover decisions -- the reference
draws 53-bit doubles) change what the optimisers DO?  Final best-f over many seeds, rng="philox" against
rng="numpy-legacy" (the reference's own stream), same shapes, deferred updating; two-sample Kolmogorov-Smirnov and
Mann-Whitney tests plus a legacy-vs-legacy split as the noise floor.  usage: rng_deviation.py [seeds]
RNG_DEV_FULL=1: the BASELINE shapes themselves (round 4; the numpy-legacy arm runs at ~10 ms per generation there)."""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy import stats
import stochopy_amd as sa

warnings.simplefilter("ignore")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 200
CASES = [
    ("de best1bin rastrigin n16 P64 300 gens (C2-like)", "de", "rastrigin", 16, {"popsize": 64, "maxiter": 300, "strategy": "best1bin"}),
    ("de rand1bin rosenbrock n16 P64 300 gens", "de", "rosenbrock", 16, {"popsize": 64, "maxiter": 300, "strategy": "rand1bin"}),
    ("pso ackley n16 P128 200 gens (C3a-like)", "pso", "ackley", 16, {"popsize": 128, "maxiter": 200}),
    ("cpso ackley n16 P128 200 gens (C3b-like)", "cpso", "ackley", 16, {"popsize": 128, "maxiter": 200}),
]


if os.environ.get("RNG_DEV_FULL") == "1":
    CASES = [
        ("de best1bin rastrigin n128 P4096 60 gens (C2)", "de", "rastrigin", 128, {"popsize": 4096, "maxiter": 60, "strategy": "best1bin"}),
        ("de best1bin rosenbrock n128 P4096 60 gens (M)", "de", "rosenbrock", 128, {"popsize": 4096, "maxiter": 60, "strategy": "best1bin"}),
        ("pso ackley n256 P16384 30 gens (C3a)", "pso", "ackley", 256, {"popsize": 16384, "maxiter": 30}),
        ("cpso ackley n256 P16384 30 gens (C3b)", "cpso", "ackley", 256, {"popsize": 16384, "maxiter": 30}),
    ]


def finals(method, obj, n, opts, rng, seeds):
    out = []
    for s in seeds:
        r = sa.optimize.minimize(getattr(sa.factory, obj), [[-5.12, 5.12]] * n, method=method,
                                 options=dict(opts, seed=int(s), rng=rng, backend="hip", updating="deferred", ftol=-1.0, xtol=0.0))
        out.append(float(r.fun))
    return np.array(out)


print(f"{S} seeds per arm; final best-f after the run; p-values of two-sample tests (small p = distributions differ)")
for label, method, obj, n, opts in CASES:
    a = finals(method, obj, n, opts, "philox", range(1000, 1000 + S))
    b = finals(method, obj, n, opts, "numpy-legacy", range(1000, 1000 + S))
    c = finals(method, obj, n, opts, "numpy-legacy", range(5000, 5000 + S))  # a second legacy sample: the noise floor
    ks, mw = stats.ks_2samp(a, b), stats.mannwhitneyu(a, b)
    ks0, mw0 = stats.ks_2samp(c, b), stats.mannwhitneyu(c, b)
    q = lambda v: " ".join("%.4g" % x for x in np.percentile(v, [10, 50, 90]))
    print(f"{label}\n   philox  p10/50/90: {q(a)}   mean {a.mean():.4g}\n   legacy  p10/50/90: {q(b)}   mean {b.mean():.4g}\n"
          f"   legacy' p10/50/90: {q(c)}   mean {c.mean():.4g}\n"
          f"   philox vs legacy : KS D={ks.statistic:.3f} p={ks.pvalue:.3f}; Mann-Whitney p={mw.pvalue:.3f}\n"
          f"   legacy' vs legacy: KS D={ks0.statistic:.3f} p={ks0.pvalue:.3f}; Mann-Whitney p={mw0.pvalue:.3f}", flush=True)
