#!/usr/bin/env python3
"""Generate golden vectors for the population hot path by RUNNING the reference.

This script is the only place that imports keurfonluu/stochopy (from
/root/reference, read-only, present in the build container only).  It writes
*data* -- configurations, inputs that cannot be regenerated from a seed, and the
reference's outputs -- into tests/golden/*.json / *.npz.  Nothing of the
reference's source travels.  The fixtures are what `oracle/` is pinned against
(tests/test_oracle_golden.py) and what the GPU parity tests compare with.

Usage (build container only):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Reference entry points exercised (paths relative to /root/reference):
    stochopy/optimize/_helpers.py:44       minimize()
    stochopy/optimize/de/_de.py:13         de.minimize
    stochopy/optimize/cpso/_cpso.py:12     cpso.minimize
    stochopy/optimize/pso/_pso.py:9        pso.minimize
    stochopy/optimize/cmaes/_cmaes.py:12   cmaes.minimize
    stochopy/optimize/na/_na.py:11         na.minimize
    stochopy/factory/benchmark.py:14-156   the seven objectives
    numpy legacy global RNG (np.random.seed/rand/permutation/randint/randn/uniform)
"""
import hashlib
import json
import os
import sys

import numpy as np

REF = os.environ.get("STOCHOPY_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

import stochopy  # noqa: E402  (the reference)
from stochopy.optimize import minimize  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def hx(a):
    """Exact, portable float encoding (C99 hex floats)."""
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        return float(a).hex()
    return [hx(v) for v in a]


def sha(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return hashlib.sha256(a.tobytes()).hexdigest()


def dump(name, obj):
    path = os.path.join(HERE, name)
    with open(path, "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name, os.path.getsize(path), "bytes")


STAMP = {
    "reference": "keurfonluu/stochopy",
    "reference_version": stochopy.__version__,
    "numpy": np.__version__,
    "generator": "tests/golden/make_golden.py",
}


# --------------------------------------------------------------------------- #
# 1. numpy legacy stream prefix, in the call order SURVEY.md section 8c lists
# --------------------------------------------------------------------------- #
def rng_stream():
    np.random.seed(42)
    calls = []

    def rec(call, val):
        val = np.asarray(val)
        if val.dtype.kind == "f":
            calls.append({"call": call, "shape": list(val.shape), "f64": hx(val.ravel()) if val.ndim else [hx(val)]})
        else:
            calls.append({"call": call, "shape": list(val.shape), "i64": [int(v) for v in val.ravel()]})

    rec("rand(3,4)", np.random.rand(3, 4))
    rec("permutation(delete(arange(8),3))", np.random.permutation(np.delete(np.arange(8), 3)))
    rec("permutation(8)", np.random.permutation(8))
    rec("randint(128,size=10)", np.random.randint(128, size=10))
    rec("randint(2)", np.random.randint(2))
    rec("randint(100,size=7)", np.random.randint(100, size=7))
    for _ in range(5):
        rec("randn(3)", np.random.randn(3))
    lo = np.array([-1.0, 0.0, 2.5])
    hi = np.array([1.0, 10.0, 2.75])
    rec("uniform(lo,hi,(4,3)) lo=[-1,0,2.5] hi=[1,10,2.75]", np.random.uniform(lo, hi, (4, 3)))
    rec("uniform(-1,1,5)", np.random.uniform(-1.0, 1.0, 5))
    rec("uniform(size=(2,3))", np.random.uniform(size=(2, 3)))
    rec("uniform(0.25,0.75)", np.random.uniform(0.25, 0.75))
    rec("normal(0,1,4)", np.random.normal(0.0, 1.0, 4))
    # position probe: the next raw 32-bit words pin how many words were consumed
    rec("randint(2**32, dtype=uint32 via randint(0,4294967296,size=4))", np.random.randint(0, 4294967296, size=4))
    out = dict(STAMP)
    out["seed"] = 42
    out["calls"] = calls
    # a longer mixed stream for the C++ host generator: seeds x lengths
    longs = []
    for seed in (0, 1, 42, 2**31 - 1, 2**32 - 1):
        np.random.seed(seed)
        d = np.random.rand(1000)
        g = np.random.randn(1001)
        p = np.random.permutation(257)
        r = np.random.randint(1000, size=500)
        longs.append(
            {
                "seed": seed,
                "rand1000_sha": sha(d),
                "rand_first": hx(d[:4]),
                "randn1001_sha": sha(g),
                "randn_first": hx(g[:4]),
                "permutation257": [int(v) for v in p],
                "randint1000x500_sum": int(r.sum()),
                "randint_first": [int(v) for v in r[:8]],
            }
        )
    out["long"] = longs
    dump("rng_stream.json", out)


# --------------------------------------------------------------------------- #
# 2. objective known answers (tests/test_factory.py values + seeded vectors)
# --------------------------------------------------------------------------- #
OBJECTIVES = ["ackley", "griewank", "quartic", "rastrigin", "rosenbrock", "sphere", "styblinski_tang"]


def factory():
    out = dict(STAMP)
    out["ones10"] = {name: hx(getattr(stochopy.factory, name)(np.ones(10))) for name in OBJECTIVES}
    # the reference's own test values (tests/test_factory.py:7-18), as data
    out["test_factory_refs"] = {
        "ackley": 3.625384938440362,
        "griewank": 0.8067591547236139,
        "quartic": 55.0,
        "rastrigin": 10.0,
        "rosenbrock": 0.0,
        "sphere": 10.0,
        "styblinski_tang": 341.6599,
    }
    cases = []
    for n in (1, 2, 3, 7, 8, 9, 10, 15, 16, 17, 64, 127, 128, 129, 130, 255, 256, 257, 511, 512, 1000, 1023, 1024, 1025, 2049):
        rs = np.random.RandomState(1000 + n)
        X = rs.uniform(-5.12, 5.12, (6, n))
        entry = {"n": n, "seed": 1000 + n, "rows": 6, "input": "RandomState(seed).uniform(-5.12,5.12,(rows,n))"}
        for name in OBJECTIVES:
            f = getattr(stochopy.factory, name)
            entry[name] = hx(np.array([f(x) for x in X]))
        cases.append(entry)
    out["cases"] = cases
    dump("factory_kat.json", out)


# --------------------------------------------------------------------------- #
# helpers to run the reference and capture a compact trace
# --------------------------------------------------------------------------- #
def run_ref(fun_name, n, method, options, x0=None, bounds=None, keep_rows=4, full=False):
    fun = getattr(stochopy.factory, fun_name)
    if bounds is None:
        bounds = [[-5.12, 5.12]] * n
    trace = []
    pops = []

    def cb(X, res):
        trace.append(float(res.fun))
        pops.append(np.array(X, dtype=np.float64, copy=True))

    opts = dict(options)
    res = minimize(fun, bounds, x0=x0, method=method, options=opts, callback=cb)
    entry = {
        "objective": fun_name,
        "ndim": n,
        "bounds": bounds if len(bounds) <= 4 else [bounds[0], "...repeated ndim times"],
        "method": method,
        "options": {k: v for k, v in options.items()},
        "x0": None if x0 is None else np.asarray(x0).tolist(),
        "result": {
            "x": hx(res.x) if full or n <= 16 else hx(res.x[:16]),
            "x_sha": sha(res.x),
            "fun": hx(res.fun),
            "nit": int(res.nit),
            "nfev": int(res.nfev),
            "status": int(res.status),
            "success": bool(res.success),
            "message": res.message,
        },
        "fun_trace": hx(np.array(trace)),
        "pop_last_sha": sha(pops[-1]),
        "pop_rows": {},
    }
    for g in sorted({0, 1, len(pops) - 1}):
        if g < len(pops):
            entry["pop_rows"][str(g)] = [hx(r[: min(n, 8)]) for r in pops[g][:keep_rows]]
    return entry, res, pops


# --------------------------------------------------------------------------- #
# 3. the reference test-suite configs that are on the hot path
#    (tests/helpers.py:13-25: 2-D Rosenbrock, maxiter 128, popsize 8, seed 42)
# --------------------------------------------------------------------------- #
def suite():
    base = {"maxiter": 128, "popsize": 8, "seed": 42, "return_all": True}
    cases = []
    arrays = {}

    def add(tag, method, extra, xref, x0=None):
        opts = dict(base)
        opts.update(extra)
        entry, res, _ = run_ref("rosenbrock", 2, method, opts, x0=x0, full=True)
        entry["tag"] = tag
        entry["xref_from_reference_tests"] = xref
        assert np.allclose(xref, res.x), (tag, xref, res.x)
        arrays[tag + "__xall"] = res.xall
        arrays[tag + "__funall"] = res.funall
        cases.append(entry)

    de_common = {"recombination": 0.1, "mutation": 0.5, "updating": "deferred"}
    add("de_rand1bin", "de", dict(de_common, strategy="rand1bin", constraints=None), [0.83228338, 0.68910339])
    add("de_rand2bin", "de", dict(de_common, strategy="rand2bin", constraints=None), [0.79409325, 0.60743767])
    add("de_best1bin", "de", dict(de_common, strategy="best1bin", constraints=None), [1.00025932, 1.00051521])
    add("de_best2bin", "de", dict(de_common, strategy="best2bin", constraints=None), [1.00515037, 1.01055037])
    add("de_rand1bin_random", "de", dict(de_common, strategy="rand1bin", constraints="Random"), [1.02340815, 1.04590782])
    pso_common = {"cognitivity": 1.49618, "sociability": 1.49618, "updating": "deferred"}
    add("pso_none", "pso", dict(pso_common, inertia=0.7298, constraints=None), [0.95990315, 0.92082304])
    add("pso_shrink", "pso", dict(pso_common, inertia=0.91, constraints="Shrink"), [0.73752093, 0.54625484])
    cpso_common = dict(pso_common, competitivity=1.0)
    add("cpso_none", "cpso", dict(cpso_common, inertia=0.7298, constraints=None), [0.95990315, 0.92082304])
    add("cpso_shrink", "cpso", dict(cpso_common, inertia=0.91, constraints="Shrink"), [0.73752093, 0.54625484])
    cma_common = {"sigma": 0.1, "muperc": 0.5}
    add("cmaes_none", "cmaes", dict(cma_common, constraints=None), [0.29967256, 0.0803311])
    add("cmaes_none_x0", "cmaes", dict(cma_common, constraints=None), [0.99998135, 0.99995618], x0=[-5.0, -5.0])
    out = dict(STAMP)
    out["cases"] = cases
    dump("suite_rosen2d.json", out)
    np.savez_compressed(os.path.join(HERE, "suite_rosen2d_xall.npz"), **arrays)
    print("wrote suite_rosen2d_xall.npz")


# --------------------------------------------------------------------------- #
# 4. BASELINE.json configs (C1..C4, M) as short traces; mid-size coverage cases
# --------------------------------------------------------------------------- #
def configs():
    out = dict(STAMP)
    cases = []
    arrays = {}

    def add(tag, *a, save_xall=False, **k):
        entry, res, pops = run_ref(*a, **k)
        entry["tag"] = tag
        cases.append(entry)
        if save_xall:
            arrays[tag + "__pops"] = np.array(pops)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status)

    # C1: README example (README.rst:93-105)
    add("C1_cmaes_readme", "rosenbrock", 2, "cmaes", {"maxiter": 100, "popsize": 10, "seed": 0}, save_xall=True)
    # M: headline shape
    add("M_de_rosen_n128_p4096", "rosenbrock", 128, "de",
        {"maxiter": 10, "popsize": 4096, "seed": 0, "updating": "deferred"})
    # C2
    add("C2_de_rastrigin_n128_p4096", "rastrigin", 128, "de",
        {"maxiter": 6, "popsize": 4096, "seed": 0, "updating": "deferred"})
    # C3a / C3b
    add("C3a_pso_ackley_n256_p16384", "ackley", 256, "pso",
        {"maxiter": 5, "popsize": 16384, "seed": 0, "updating": "deferred"})
    add("C3b_cpso_ackley_n256_p16384", "ackley", 256, "cpso",
        {"maxiter": 5, "popsize": 16384, "seed": 0, "updating": "deferred"})
    # C4
    add("C4_cmaes_rosen_n512_p1024", "rosenbrock", 512, "cmaes", {"maxiter": 5, "popsize": 1024, "seed": 0})

    # mid-size coverage: every DE strategy x constraint, ragged n (not multiples of 8)
    for strat in ("rand1bin", "rand2bin", "best1bin", "best2bin"):
        for cons in (None, "Random"):
            for (n, P) in ((13, 40), (130, 96)):
                add(f"de_{strat}_{cons}_n{n}_p{P}", "rosenbrock", n, "de",
                    {"maxiter": 12, "popsize": P, "seed": 7, "updating": "deferred", "strategy": strat,
                     "constraints": cons, "mutation": 0.6, "recombination": 0.8}, save_xall=(n == 13))
    # narrow bounds so that Random actually resamples
    add("de_rand1bin_Random_tight_n16_p64", "sphere", 16, "de",
        {"maxiter": 15, "popsize": 64, "seed": 3, "updating": "deferred", "strategy": "rand1bin",
         "constraints": "Random", "mutation": 1.5, "recombination": 0.9},
        bounds=[[-1.0, 2.0]] * 16, save_xall=True)
    # PSO / CPSO with and without Shrink; CPSO restarts fire at P=256 n=16
    for method in ("pso", "cpso"):
        for cons in (None, "Shrink"):
            for fun in ("ackley", "rosenbrock"):
                add(f"{method}_{cons}_{fun}_n16_p256", fun, 16, method,
                    {"maxiter": 30, "popsize": 256, "seed": 5, "updating": "deferred", "constraints": cons,
                     "inertia": 0.7298, "cognitivity": 1.49618, "sociability": 1.49618}, save_xall=(fun == "rosenbrock" and method == "cpso" and cons == "Shrink"))
    add("cpso_None_sphere_n37_p100_g08", "sphere", 37, "cpso",
        {"maxiter": 25, "popsize": 100, "seed": 11, "updating": "deferred", "competitivity": 0.8})
    # CMA-ES mid sizes (status other than -1 as well)
    add("cmaes_rosen_n20_p20", "rosenbrock", 20, "cmaes", {"maxiter": 300, "popsize": 20, "seed": 0})
    add("cmaes_sphere_n8_p12", "sphere", 8, "cmaes", {"maxiter": 400, "popsize": 12, "seed": 1})
    add("cmaes_rastrigin_n33_p64", "rastrigin", 33, "cmaes", {"maxiter": 40, "popsize": 64, "seed": 2, "sigma": 0.3})
    # early termination by ftol (status 1) / xtol (status 0) for DE and PSO
    add("de_status_sphere_n4_p32", "sphere", 4, "de",
        {"maxiter": 400, "popsize": 32, "seed": 9, "updating": "deferred", "ftol": 1e-6, "xtol": 1e-3})
    add("pso_status_sphere_n4_p32", "sphere", 4, "pso",
        {"maxiter": 400, "popsize": 32, "seed": 9, "updating": "deferred", "ftol": 1e-6, "xtol": 1e-3})
    out["cases"] = cases
    dump("configs.json", out)
    np.savez_compressed(os.path.join(HERE, "configs_pops.npz"), **arrays)
    print("wrote configs_pops.npz", os.path.getsize(os.path.join(HERE, "configs_pops.npz")))


# --------------------------------------------------------------------------- #
# 4b. BASELINE configs at FULL size over longer runs (round 4): best-f of every generation, the final x, a projection
#     of the WHOLE population every few generations (P numbers per look: X @ w with a fixed seeded w -- a flipped `<`
#     anywhere in the population moves one of them by O(search range)), and for CPSO the competitive restart's
#     bookkeeping (cpso/_cpso.py:405-426): for every generation in which it fires, nw and the rows it re-seeds.
# --------------------------------------------------------------------------- #
def configs_long():
    cpso_mod = sys.modules["stochopy.optimize.cpso._cpso"]  # (the package re-exports `minimize` under the module's name)

    out = dict(STAMP)
    cases = []
    arrays = {}

    def add(tag, fun_name, n, method, options, every):
        fun = getattr(stochopy.factory, fun_name)
        bounds = [[-5.12, 5.12]] * n
        w = np.random.RandomState(20240929).standard_normal(n)
        trace, looks, proj, rows8 = [], [], [], {}
        restarts = []
        gen = [0]

        def cb(X, res):
            gen[0] += 1
            trace.append(float(res.fun))
            if gen[0] == 1 or gen[0] % every == 0 or gen[0] == options["maxiter"]:
                looks.append(gen[0])
                proj.append(np.asarray(X, dtype=np.float64) @ w)
                rows8[str(gen[0])] = [hx(r[:8]) for r in X[:4]]

        orig = cpso_mod.restart

        def spy(it, X, V, pbest, gbest, pbestfit, *a):
            before = pbestfit.copy()
            ret = orig(it, X, V, pbest, gbest, pbestfit, *a)
            rows = np.flatnonzero((ret[3] == 1.0e30) & (before != 1.0e30))
            if len(rows):
                restarts.append((int(it), rows.astype(np.int32)))
            return ret

        cpso_mod.restart = spy
        try:
            res = minimize(fun, bounds, method=method, options=dict(options), callback=cb)
        finally:
            cpso_mod.restart = orig
        entry = {
            "tag": tag, "objective": fun_name, "ndim": n, "bounds": [bounds[0], "...repeated ndim times"],
            "method": method, "options": dict(options), "x0": None,
            "result": {"x": hx(res.x), "fun": hx(res.fun), "nit": int(res.nit), "nfev": int(res.nfev),
                       "status": int(res.status), "success": bool(res.success), "message": res.message},
            "fun_trace": hx(np.array(trace)),
            "looks": looks,                      # generations (1-based callback count) whose population is projected
            "pop_rows": rows8,                   # first 8 elements of the first 4 rows at those generations
            "restarts": [[it, int(len(r))] for it, r in restarts],
        }
        arrays[tag + "__w"] = w
        arrays[tag + "__proj"] = np.array(proj)
        for it, r in restarts:
            arrays[tag + "__restart_%d" % it] = np.sort(r)
        cases.append(entry)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status, "restarts", entry["restarts"][:6], flush=True)

    add("C2L_de_rastrigin_n128_p4096", "rastrigin", 128, "de",
        {"maxiter": 40, "popsize": 4096, "seed": 0, "updating": "deferred"}, 5)
    add("C3aL_pso_ackley_n256_p16384", "ackley", 256, "pso",
        {"maxiter": 30, "popsize": 16384, "seed": 0, "updating": "deferred"}, 5)
    add("C3bL_cpso_ackley_n256_p16384", "ackley", 256, "cpso",
        {"maxiter": 30, "popsize": 16384, "seed": 0, "updating": "deferred"}, 5)
    add("C4L_cmaes_rosen_n512_p1024", "rosenbrock", 512, "cmaes", {"maxiter": 16, "popsize": 1024, "seed": 0}, 1)
    out["cases"] = cases
    dump("configs_long.json", out)
    np.savez_compressed(os.path.join(HERE, "configs_long.npz"), **arrays)
    print("wrote configs_long.npz", os.path.getsize(os.path.join(HERE, "configs_long.npz")))


# --------------------------------------------------------------------------- #
# 5. CMA-ES with constraints="Penalize" (stochopy/optimize/cmaes/_constraints.py:4-82): the reference's own
#    test rows (tests/test_optimize.py:9-20) and boxes in which the penalty is actually at work
# --------------------------------------------------------------------------- #
def penalize():
    out = dict(STAMP)
    cases = []
    arrays = {}

    def add(tag, fun, n, opts, bounds=None, x0=None, xref=None):
        o = dict(opts, constraints="Penalize", return_all=True)
        entry, res, pops = run_ref(fun, n, "cmaes", o, x0=x0, bounds=bounds, full=True)
        entry["tag"] = tag
        if xref is not None:
            entry["xref_from_reference_tests"] = xref
            assert np.allclose(xref, res.x), (tag, xref, res.x)
        arrays[tag + "__xall"] = res.xall
        arrays[tag + "__funall"] = res.funall
        cases.append(entry)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status)

    suite_opts = {"maxiter": 128, "popsize": 8, "seed": 42, "sigma": 0.1, "muperc": 0.5}
    add("cmaes_penalize", "rosenbrock", 2, suite_opts, xref=[0.18765786, 0.05858025])
    add("cmaes_penalize_x0", "rosenbrock", 2, suite_opts, x0=[-5.0, -5.0], xref=[0.99998135, 0.99995618])
    # optimum outside / on the edge of the box: the mean leaves the box and the boundary weights grow
    add("cmaes_penalize_rosen_n4_edge", "rosenbrock", 4, {"maxiter": 60, "popsize": 12, "seed": 3},
        bounds=[[-5.12, 0.5]] * 4)
    add("cmaes_penalize_sphere_n6_outside", "sphere", 6, {"maxiter": 80, "popsize": 10, "seed": 7, "sigma": 0.3},
        bounds=[[1.0, 5.0]] * 6)
    add("cmaes_penalize_rastrigin_n24_p48", "rastrigin", 24, {"maxiter": 40, "popsize": 48, "seed": 1, "sigma": 0.5},
        bounds=[[-5.12, 2.0]] * 24)
    out["cases"] = cases
    dump("cmaes_penalize.json", out)
    np.savez_compressed(os.path.join(HERE, "cmaes_penalize_xall.npz"), **arrays)
    print("wrote cmaes_penalize_xall.npz", os.path.getsize(os.path.join(HERE, "cmaes_penalize_xall.npz")))


# --------------------------------------------------------------------------- #
# 6. VD-CMA (stochopy/optimize/vdcma/_vdcma.py:12-460): the reference's own test rows
#    (tests/test_optimize.py:119-132) and mid-size problems where v and d actually adapt (n > 5)
# --------------------------------------------------------------------------- #
def vdcma():
    out = dict(STAMP)
    cases = []
    arrays = {}

    def add(tag, fun, n, opts, bounds=None, x0=None, xref=None):
        o = dict(opts, return_all=True)
        entry, res, pops = run_ref(fun, n, "vdcma", o, x0=x0, bounds=bounds, full=True)
        entry["tag"] = tag
        if xref is not None:
            entry["xref_from_reference_tests"] = xref
            assert np.allclose(xref, res.x), (tag, xref, res.x)
        arrays[tag + "__xall"] = res.xall
        arrays[tag + "__funall"] = res.funall
        cases.append(entry)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status)

    suite_opts = {"maxiter": 128, "popsize": 8, "seed": 42, "sigma": 0.1, "muperc": 0.5}
    add("vdcma_none", "rosenbrock", 2, dict(suite_opts, constraints=None), xref=[0.90013445, 0.85037782])
    add("vdcma_none_x0", "rosenbrock", 2, dict(suite_opts, constraints=None), x0=[-5.0, -5.0],
        xref=[0.84059993, 0.69998341])
    add("vdcma_penalize", "rosenbrock", 2, dict(suite_opts, constraints="Penalize"), xref=[0.90013445, 0.85037782])
    add("vdcma_penalize_x0", "rosenbrock", 2, dict(suite_opts, constraints="Penalize"), x0=[-5.0, -5.0],
        xref=[0.82405114, 0.61993136])
    add("vdcma_rosen_n12_p16", "rosenbrock", 12, {"maxiter": 150, "popsize": 16, "seed": 3})
    add("vdcma_sphere_n9_outside", "sphere", 9,
        {"maxiter": 80, "popsize": 12, "seed": 7, "sigma": 0.3, "constraints": "Penalize"}, bounds=[[1.0, 5.0]] * 9)
    add("vdcma_rastrigin_n40_p24", "rastrigin", 40, {"maxiter": 60, "popsize": 24, "seed": 1, "sigma": 0.4})
    add("vdcma_sphere_n16_ftol", "sphere", 16, {"maxiter": 600, "popsize": 16, "seed": 5, "ftol": 1e-6})
    out["cases"] = cases
    dump("vdcma.json", out)
    np.savez_compressed(os.path.join(HERE, "vdcma_xall.npz"), **arrays)
    print("wrote vdcma_xall.npz", os.path.getsize(os.path.join(HERE, "vdcma_xall.npz")))


# --------------------------------------------------------------------------- #
# 6b. WIDE rows (n > 4096): the reference has no dimension limit (de/_de.py:208-218), and VD-CMA exists for long vectors
#     (vdcma/_vdcma.py:144-458).  One VD-CMA run at n = 8192, objective values of long rows (numpy's pairwise sums with
#     hundreds of leaves), and short DE / PSO / CPSO runs at n = 4097 -- the first length the wide kernels serve.
# --------------------------------------------------------------------------- #
def vdcma_wide():
    out = dict(STAMP)
    arrays = {}
    cases = []
    n = 8192
    o = {"maxiter": 12, "popsize": 16, "seed": 9, "sigma": 0.3}
    entry, res, pops = run_ref("rosenbrock", n, "vdcma", o, bounds=[[-3.0, 3.0]] * n)
    entry["tag"] = "vdcma_wide_rosen_n8192_p16"
    arrays[entry["tag"] + "__x"] = res.x
    cases.append(entry)
    print(" ", entry["tag"], "fun", float(res.fun), "nit", res.nit, "status", res.status)
    for tag, method, n, o in (
        ("de_wide_rosen_n4097_p12", "de", 4097, {"maxiter": 6, "popsize": 12, "seed": 2, "strategy": "best1bin", "updating": "deferred"}),
        ("de_wide_sphere_n9000_p10_random", "de", 9000, {"maxiter": 5, "popsize": 10, "seed": 4, "strategy": "rand1bin",
                                                         "constraints": "Random", "updating": "deferred"}),
        ("pso_wide_rosen_n4097_p12", "pso", 4097, {"maxiter": 6, "popsize": 12, "seed": 2, "updating": "deferred"}),
        ("cpso_wide_sphere_n5000_p12_shrink", "cpso", 5000, {"maxiter": 8, "popsize": 12, "seed": 6, "constraints": "Shrink",
                                                             "competitivity": 1.0, "updating": "deferred"}),
    ):
        fun = tag.split("_")[2].replace("rosen", "rosenbrock")
        entry, res, pops = run_ref(fun, n, method, o, bounds=[[-2.0, 2.0]] * n)
        entry["tag"] = tag
        arrays[tag + "__x"] = res.x
        arrays[tag + "__pop_last"] = pops[-1]
        cases.append(entry)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status)
    # objective values of long rows straight from the reference's functions
    rs = np.random.RandomState(20260929)
    kat = {}
    for n in (4097, 8192, 20001, 65536):
        X = rs.uniform(-5.12, 5.12, (3, n))
        arrays["kat_X_%d" % n] = X
        for name in ("ackley", "griewank", "quartic", "rastrigin", "rosenbrock", "sphere", "styblinski_tang"):
            kat["%s_%d" % (name, n)] = hx(np.array([getattr(stochopy.factory, name)(x) for x in X]))
    out["cases"] = cases
    out["objective_kat"] = kat
    dump("vdcma_wide.json", out)
    np.savez_compressed(os.path.join(HERE, "vdcma_wide.npz"), **arrays)
    print("wrote vdcma_wide.npz", os.path.getsize(os.path.join(HERE, "vdcma_wide.npz")))


# --------------------------------------------------------------------------- #
# 7. updating="immediate" (de/_de.py:354-391 de_async, cpso/_cpso.py:364-402 pso_async,
#    _common.py:163-194 selection_async): the reference's own test rows (tests/test_optimize.py:27-117) and
#    mid-size problems for every strategy / constraint, incl. runs that stop on ftol / xtol
# --------------------------------------------------------------------------- #
def immediate():
    out = dict(STAMP)
    cases = []
    arrays = {}

    def add(tag, fun, n, method, opts, xref=None):
        o = dict(opts, updating="immediate", return_all=True)
        entry, res, pops = run_ref(fun, n, method, o, full=True)
        entry["tag"] = tag
        if xref is not None:
            entry["xref_from_reference_tests"] = xref
            assert np.allclose(xref, res.x), (tag, xref, res.x)
        arrays[tag + "__xall"] = res.xall
        arrays[tag + "__funall"] = res.funall
        cases.append(entry)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status)

    base = {"maxiter": 128, "popsize": 8, "seed": 42}
    de_common = dict(base, recombination=0.1, mutation=0.5)
    add("de_rand1bin_immediate", "rosenbrock", 2, "de", dict(de_common, strategy="rand1bin", constraints=None),
        [0.85658185, 0.726094])
    add("de_rand1bin_random_immediate", "rosenbrock", 2, "de", dict(de_common, strategy="rand1bin", constraints="Random"),
        [0.99438151, 0.9944796])
    pso_common = dict(base, cognitivity=1.49618, sociability=1.49618)
    add("pso_none_immediate", "rosenbrock", 2, "pso", dict(pso_common, inertia=0.7298, constraints=None),
        [0.95909508, 0.91977272])
    add("pso_shrink_immediate", "rosenbrock", 2, "pso", dict(pso_common, inertia=0.91, constraints="Shrink"),
        [0.76668308, 0.58381385])
    cpso_common = dict(pso_common, competitivity=1.0)
    add("cpso_none_immediate", "rosenbrock", 2, "cpso", dict(cpso_common, inertia=0.7298, constraints=None),
        [0.93258856, 0.86919435])
    add("cpso_shrink_immediate", "rosenbrock", 2, "cpso", dict(cpso_common, inertia=0.91, constraints="Shrink"),
        [0.76668308, 0.58381385])
    for strat in ("rand1bin", "rand2bin", "best1bin", "best2bin"):
        for cons in (None, "Random"):
            add("de_%s_%s_n10_p40_immediate" % (strat, cons or "none"), "rastrigin", 10, "de",
                {"maxiter": 12, "popsize": 40, "seed": 7, "strategy": strat, "constraints": cons,
                 "mutation": 0.9 if cons else 0.5})
    add("de_best1bin_rosen_n150_p24_immediate", "rosenbrock", 150, "de", {"maxiter": 8, "popsize": 24, "seed": 2})
    for cons in (None, "Shrink"):
        add("pso_%s_ackley_n16_p64_immediate" % (cons or "none"), "ackley", 16, "pso",
            {"maxiter": 30, "popsize": 64, "seed": 5, "constraints": cons, "inertia": 0.91 if cons else 0.7298})
        add("cpso_%s_ackley_n16_p96_immediate" % (cons or "none"), "ackley", 16, "cpso",
            {"maxiter": 30, "popsize": 96, "seed": 5, "constraints": cons, "inertia": 0.91 if cons else 0.7298})
    add("pso_styblinski_n130_p20_immediate", "styblinski_tang", 130, "pso", {"maxiter": 10, "popsize": 20, "seed": 4})
    add("de_status_sphere_n4_p32_immediate", "sphere", 4, "de",
        {"maxiter": 400, "popsize": 32, "seed": 9, "ftol": 1e-6, "xtol": 1e-3})
    add("pso_status_sphere_n4_p32_immediate", "sphere", 4, "pso",
        {"maxiter": 400, "popsize": 32, "seed": 9, "ftol": 1e-6, "xtol": 1e-3})
    out["cases"] = cases
    dump("immediate.json", out)
    np.savez_compressed(os.path.join(HERE, "immediate_xall.npz"), **arrays)
    print("wrote immediate_xall.npz", os.path.getsize(os.path.join(HERE, "immediate_xall.npz")))


# --------------------------------------------------------------------------- #
# 8. Neighbourhood Algorithm (na/_na.py:131-305): the reference's own test row (tests/test_optimize.py:89-92) and
#    coverage cases (more axes, x0, a fixed axis, early stop, history options)
# --------------------------------------------------------------------------- #
def na():
    out = dict(STAMP)
    cases = []
    arrays = {}

    def add(tag, fun, n, opts, bounds=None, x0=None, xref=None):
        o = dict(opts, return_all=True)
        entry, res, pops = run_ref(fun, n, "na", o, x0=x0, bounds=bounds, full=True)
        entry["tag"] = tag
        if xref is not None:
            entry["xref_from_reference_tests"] = xref
            assert np.allclose(xref, res.x), (tag, xref, res.x)
        arrays[tag + "__xall"] = res.xall
        arrays[tag + "__funall"] = res.funall
        cases.append(entry)
        print(" ", tag, "fun", float(res.fun), "nit", res.nit, "status", res.status)

    suite_opts = {"maxiter": 128, "popsize": 8, "seed": 42, "nrperc": 0.5}
    add("na_suite", "rosenbrock", 2, suite_opts, xref=[1.14849912, 1.31885465])
    rs = np.random.RandomState(11)
    add("na_suite_x0", "rosenbrock", 2, suite_opts, x0=rs.uniform(-5.12, 5.12, (8, 2)))
    add("na_sphere_n5_p12", "sphere", 5, {"maxiter": 40, "popsize": 12, "seed": 3, "nrperc": 0.25})
    add("na_rastrigin_n3_p20", "rastrigin", 3, {"maxiter": 30, "popsize": 20, "seed": 9, "nrperc": 1.0})
    add("na_rosen_n12_p16", "rosenbrock", 12, {"maxiter": 20, "popsize": 16, "seed": 5, "nrperc": 0.5})
    add("na_fixed_axis", "sphere", 4, {"maxiter": 25, "popsize": 10, "seed": 2, "nrperc": 0.3},
        bounds=[[-5.12, 5.12], [1.5, 1.5], [-2.0, 3.0], [-5.12, 5.12]])
    add("na_sphere_ftol", "sphere", 2, {"maxiter": 400, "popsize": 16, "seed": 7, "nrperc": 0.5, "ftol": 1e-4})
    add("na_best_only_history", "rosenbrock", 3, {"maxiter": 25, "popsize": 9, "seed": 4, "nrperc": 0.4, "verbosity": 0.0})
    add("na_half_history", "ackley", 4, {"maxiter": 20, "popsize": 10, "seed": 6, "nrperc": 0.5, "verbosity": 0.5})
    out["cases"] = cases
    dump("na.json", out)
    np.savez_compressed(os.path.join(HERE, "na_xall.npz"), **arrays)
    print("wrote na_xall.npz", os.path.getsize(os.path.join(HERE, "na_xall.npz")))


if __name__ == "__main__":
    which = sys.argv[1:] or ["rng", "factory", "suite", "configs", "penalize", "vdcma", "immediate", "na"]
    if "na" in which:
        na()
    if "immediate" in which:
        immediate()
    if "vdcma" in which:
        vdcma()
    if "vdcma_wide" in which:
        vdcma_wide()
    if "penalize" in which:
        penalize()
    if "rng" in which:
        rng_stream()
    if "factory" in which:
        factory()
    if "suite" in which:
        suite()
    if "configs" in which:
        configs()
    if "configs_long" in which:
        configs_long()
