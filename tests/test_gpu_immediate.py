"""GPU parity for updating="immediate" (strict_updating=True): the sequential sweeps of csrc/sx_async.hip
through the C ABI vs the vectors captured from the reference (numpy-legacy stream) and vs the oracle (Philox).

Mirrors the reference's immediate rows in tests/test_optimize.py:27-117 (helpers.optimize_parallel: allclose on
x, bounds respected with return_all)."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, case_bounds, load_golden, unhex

pytestmark = pytest.mark.gpu

EXACT = {"rosenbrock", "sphere"}  # + - * only: numpy's bits
CASES = load_golden("immediate.json")["cases"]


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


def _run_hip(sa, case, rng="numpy-legacy", **extra):
    trace = []
    opts = dict(case["options"])
    opts.update({"backend": "hip", "rng": rng, "strict_updating": True})
    opts.update(extra)
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), case_bounds(case), x0=case["x0"],
                               method=case["method"], options=opts,
                               callback=lambda X, r: trace.append((float(r.fun), X.copy())))
    return res, trace


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["tag"])
def test_immediate_matches_reference_golden(sa, case):
    """Same seed => the reference's run: per-generation best-f, x, xall / funall, nit, status."""
    res, trace = _run_hip(sa, case)
    ref = case["result"]
    arrays = np.load(os.path.join(GOLDEN, "immediate_xall.npz"))
    got = np.array([t[0] for t in trace])
    want = unhex(case["fun_trace"])
    if case["objective"] in EXACT:
        assert np.array_equal(got, want)
        assert np.array_equal(unhex(ref["x"]), res.x) and float(res.fun).hex() == ref["fun"]
        assert np.array_equal(arrays[case["tag"] + "__xall"], res.xall)
        assert np.array_equal(arrays[case["tag"] + "__funall"], res.funall)
    else:
        assert np.allclose(got, want, rtol=1e-6, atol=0)  # north-star tolerance: best-f within 1e-6 rel
        assert np.allclose(arrays[case["tag"] + "__funall"], res.funall, rtol=1e-6, atol=1e-9)
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)  # tests/helpers.py:22
    if case["options"].get("constraints"):
        lo, hi = np.transpose(case_bounds(case))
        assert np.all(res.xall + 1.0e-15 >= lo) and np.all(res.xall - 1.0e-15 <= hi)  # tests/helpers.py:23-25


PHILOX = [
    ("de", "rosenbrock", 2, {"popsize": 8, "maxiter": 40, "strategy": "rand1bin"}),
    ("de", "rosenbrock", 24, {"popsize": 50, "maxiter": 15, "strategy": "best1bin"}),
    ("de", "sphere", 70, {"popsize": 33, "maxiter": 12, "strategy": "rand2bin", "constraints": "Random", "mutation": 1.5}),
    ("de", "rosenbrock", 128, {"popsize": 64, "maxiter": 6, "strategy": "best2bin"}),
    ("de", "sphere", 300, {"popsize": 20, "maxiter": 6, "strategy": "best1bin", "constraints": "Random", "mutation": 1.2}),
    ("de", "rosenbrock", 1000, {"popsize": 12, "maxiter": 4, "strategy": "rand1bin"}),
    ("de", "sphere", 4, {"popsize": 32, "maxiter": 400, "ftol": 1e-6, "xtol": 1e-3}),
    ("pso", "rosenbrock", 2, {"popsize": 8, "maxiter": 40}),
    ("pso", "sphere", 40, {"popsize": 64, "maxiter": 20, "constraints": "Shrink", "inertia": 0.91}),
    ("pso", "rosenbrock", 130, {"popsize": 30, "maxiter": 8}),
    ("pso", "sphere", 520, {"popsize": 16, "maxiter": 6, "constraints": "Shrink", "inertia": 0.95}),
    ("pso", "sphere", 4, {"popsize": 32, "maxiter": 300, "ftol": 1e-6, "xtol": 1e-3}),
    ("cpso", "sphere", 16, {"popsize": 96, "maxiter": 30, "constraints": "Shrink", "inertia": 0.91}),
    ("cpso", "rosenbrock", 8, {"popsize": 128, "maxiter": 40}),
    # many rounds of 64 / 32 / 16 individuals per sweep, ragged last round
    ("de", "rosenbrock", 16, {"popsize": 701, "maxiter": 6, "strategy": "best1bin"}),
    ("de", "sphere", 100, {"popsize": 333, "maxiter": 5, "strategy": "rand1bin", "constraints": "Random", "mutation": 1.3}),
    ("de", "rosenbrock", 300, {"popsize": 150, "maxiter": 4, "strategy": "rand2bin"}),
    ("pso", "sphere", 20, {"popsize": 517, "maxiter": 8, "constraints": "Shrink", "inertia": 0.91}),
    ("cpso", "rosenbrock", 200, {"popsize": 130, "maxiter": 8}),
]


@pytest.mark.parametrize("method,objective,n,opts", PHILOX, ids=lambda v: str(v) if not isinstance(v, dict) else "")
def test_immediate_philox_vs_oracle(sa, method, objective, n, opts):
    """In-kernel Philox draws (same counters as the synchronous kernels): the sweep vs the oracle's, bit for bit
    (+ - * objectives), incl. return_all."""
    bounds = [[-5.12, 5.12]] * n
    o = dict(opts, seed=11, updating="immediate", return_all=True)
    ref = oracle.minimize(objective, bounds, method=method, options=dict(o), rng="philox")
    res = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method=method,
                               options=dict(o, backend="hip", rng="philox", strict_updating=True))
    assert (res.nit, res.nfev, res.status) == (ref["nit"], ref["nfev"], ref["status"])
    assert np.array_equal(res.x, ref["x"]) and res.fun == ref["fun"]
    assert np.array_equal(res.xall, ref["xall"]) and np.array_equal(res.funall, ref["funall"])


def test_immediate_is_honoured_by_default(sa):
    """The reference runs de_async for a default call (updating="immediate", no parallel backend; de/_de.py:142-145).
    So does this backend when it can (one GPU, factory objective): same seed, the reference's result.
    strict_updating=False is the explicit throughput choice (deferred, silently); a run that cannot be an ordered
    sweep (caller-supplied objective) is deferred WITH a warning."""
    import warnings

    bounds = [[-5.12, 5.12]] * 2
    o = {"popsize": 8, "maxiter": 30, "seed": 42, "backend": "hip"}
    a = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o))  # default: immediate
    b = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o, updating="deferred"))
    c = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de",
                             options=dict(o, updating="immediate", strict_updating=True))
    d = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o, strict_updating=False))
    assert np.array_equal(a.x, c.x) and np.array_equal(d.x, b.x) and not np.array_equal(a.x, b.x)
    ref = oracle.minimize("rosenbrock", bounds, method="de",
                          options={"popsize": 8, "maxiter": 30, "seed": 42, "updating": "immediate"})  # the reference's default
    assert np.array_equal(a.x, ref.x) and a.fun == ref.fun and a.nit == ref.nit
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        e = sa.optimize.minimize(sa.factory.batched(lambda X: (X * X).sum(dim=1)), bounds, method="de", options=dict(o))
    assert any("deferred" in str(w.message) for w in seen)
    f = sa.optimize.minimize(sa.factory.sphere, bounds, method="de", options=dict(o, updating="deferred"))
    assert np.allclose(e.x, f.x) and np.isclose(e.fun, f.fun)


def test_immediate_pso_updates_x0_in_place(sa):
    """pso_async assigns X[i] row by row: the caller's x0 is the working swarm (cpso/_cpso.py:389)."""
    rs = np.random.RandomState(3)
    x0 = rs.uniform(-5.12, 5.12, (12, 3))
    keep = x0.copy()
    sa.optimize.minimize(sa.factory.sphere, [[-5.12, 5.12]] * 3, x0=x0, method="pso",
                         options={"popsize": 12, "maxiter": 5, "seed": 1, "backend": "hip", "strict_updating": True})
    assert not np.array_equal(x0, keep)
