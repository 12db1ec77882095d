import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def load_golden(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def unhex(v):
    if isinstance(v, str):
        return float.fromhex(v)
    return np.array([unhex(x) for x in v], dtype=np.float64)


def case_bounds(case):
    n = case["ndim"]
    b = case["bounds"]
    if len(b) == n and not isinstance(b[-1], str):
        return b
    return [b[0]] * n


@pytest.fixture(scope="session")
def golden_configs():
    return {c["tag"]: c for c in load_golden("configs.json")["cases"]}


@pytest.fixture(scope="session")
def golden_suite():
    return {c["tag"]: c for c in load_golden("suite_rosen2d.json")["cases"]}


def check_long_case(sa, case, x_tol=1e-6, proj_tol=1e-6, restart_rows=None):
    """BASELINE config at full size over a longer run against the vectors captured from the reference
    (tests/golden/configs_long.*, numpy-legacy draws, same seed): best-f of EVERY generation within 1e-6 rel (the
    north-star tolerance), nit / status, the final x within `x_tol` of the search range, and the projection X @ w of the
    WHOLE population every few generations within `proj_tol` of the search range -- a `<` decision that went the other way
    anywhere in the population moves that row's entry by O(search range).  `restart_rows`: a dict the caller fills
    (generation -> sorted rows the competitive restart re-seeded); compared exactly."""
    arrays = np.load(os.path.join(GOLDEN, "configs_long.npz"))
    tag = case["tag"]
    w = arrays[tag + "__w"]
    looks, proj, rows8, count, trace = set(case["looks"]), [], {}, [0], []

    def cb(X, r):
        count[0] += 1
        trace.append(float(r.fun))
        if count[0] in looks:
            proj.append(np.asarray(X) @ w)
            rows8[str(count[0])] = np.array(X[:4, :8], copy=True)

    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), case_bounds(case), method=case["method"],
                               options=opts, callback=cb)
    ref = case["result"]
    rng_width = 10.24
    assert np.allclose(np.array(trace), unhex(case["fun_trace"]), rtol=1e-6, atol=0)
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    assert np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=0)
    assert np.allclose(res.x, unhex(ref["x"]), rtol=0, atol=x_tol * rng_width)
    want = arrays[tag + "__proj"]
    assert len(proj) == len(want)
    for g, a, b in zip(case["looks"], proj, want):
        bad = np.flatnonzero(~np.isclose(a, b, rtol=0, atol=proj_tol * rng_width))
        assert bad.size == 0, f"{tag}: generation {g}: rows {bad[:8]} of the population differ from the reference's"
    for g, rows in case["pop_rows"].items():
        assert np.allclose(unhex(rows), rows8[g], rtol=0, atol=x_tol * rng_width)
    if restart_rows is not None:
        assert [[it, len(r)] for it, r in sorted(restart_rows.items())] == case["restarts"]
        for it, r in restart_rows.items():
            assert np.array_equal(arrays[tag + "__restart_%d" % it], r), f"{tag}: restart of generation {it}"
    return res
