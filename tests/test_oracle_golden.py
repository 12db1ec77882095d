"""Pin the CPU oracle (oracle/) against vectors captured from the reference itself.

CPU-only.  Bit-for-bit: the oracle consumes numpy's legacy stream in the
reference's order and uses the same numpy arithmetic, so every per-generation
best-f, the final x, nit and status must be IDENTICAL to what
tests/golden/make_golden.py recorded from keurfonluu/stochopy v2.3.0.
Mirrors the reference's tests/helpers.py:13-25 (golden xref) and
tests/test_factory.py:7-23 (objective known answers).
"""
import hashlib
import os

import numpy as np
import pytest

import oracle
from oracle.objectives import OBJECTIVES, pairwise_sum_py
from conftest import GOLDEN, case_bounds, load_golden, unhex

CONFIGS = load_golden("configs.json")["cases"]
SUITE = load_golden("suite_rosen2d.json")["cases"]
FACTORY = load_golden("factory_kat.json")


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_objective_known_answers(name):
    """tests/test_factory.py:7-23: value at ones(10), np.allclose."""
    f = OBJECTIVES[name](np.ones(10))[0]
    assert np.allclose(FACTORY["test_factory_refs"][name], f)
    assert float(f).hex() == FACTORY["ones10"][name]


@pytest.mark.parametrize("case", FACTORY["cases"], ids=lambda c: "n%d" % c["n"])
def test_objectives_bit_exact(case):
    X = np.random.RandomState(case["seed"]).uniform(-5.12, 5.12, (case["rows"], case["n"]))
    for name, f in OBJECTIVES.items():
        assert np.array_equal(unhex(case[name]), f(X)), name


@pytest.mark.parametrize("m", [0, 1, 5, 7, 8, 9, 15, 16, 17, 127, 128, 129, 135, 255, 256, 1023, 1024, 1031])
def test_pairwise_restatement_matches_numpy(m):
    """the summation order the device kernels implement."""
    a = np.random.RandomState(m).uniform(-1e3, 1e3, m)
    assert pairwise_sum_py(a) == a.sum()


def _run(case, trace):
    return oracle.minimize(case["objective"], case_bounds(case), x0=case["x0"], method=case["method"],
                           options=dict(case["options"]), callback=lambda X, r: trace.append((float(r.fun), X.copy())))


@pytest.mark.parametrize("case", CONFIGS, ids=lambda c: c["tag"])
def test_configs_bit_exact(case):
    trace = []
    res = _run(case, trace)
    ref = case["result"]
    assert np.array_equal(unhex(case["fun_trace"]), np.array([t[0] for t in trace]))
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    assert float(res.fun).hex() == ref["fun"]
    assert _sha(res.x) == ref["x_sha"]
    assert _sha(trace[-1][1]) == case["pop_last_sha"]
    for g, rows in case["pop_rows"].items():
        got = trace[int(g)][1]
        for r, row in enumerate(rows):
            w = len(row)
            assert np.array_equal(unhex(row), got[r, :w])


@pytest.mark.parametrize("case", SUITE, ids=lambda c: c["tag"])
def test_reference_suite_xrefs(case):
    """The reference's own golden xrefs (tests/test_optimize.py:9-118), hot-path rows only."""
    trace = []
    res = _run(case, trace)
    assert np.allclose(case["xref_from_reference_tests"], res.x)  # tests/helpers.py:22
    assert np.array_equal(unhex(case["result"]["x"]), res.x)
    arrays = np.load(os.path.join(GOLDEN, "suite_rosen2d_xall.npz"))
    assert np.array_equal(arrays[case["tag"] + "__xall"], res.xall)
    assert np.array_equal(arrays[case["tag"] + "__funall"], res.funall)
    if case["options"].get("constraints"):
        assert np.all(res.xall + 1.0e-15 >= -5.12) and np.all(res.xall - 1.0e-15 <= 5.12)  # helpers.py:23-25


LONG = load_golden("configs_long.json")["cases"]


def long_arrays():
    return np.load(os.path.join(GOLDEN, "configs_long.npz"))


@pytest.mark.parametrize("case", LONG, ids=lambda c: c["tag"])
def test_long_configs_bit_exact(case):
    """BASELINE configs 2 / 3a / 3b / 4 at full size over 40 / 30 / 30 / 16 generations (round 4): best-f of every
    generation, the final x, a projection of the WHOLE population every few generations, and for CPSO the restart's
    bookkeeping (cpso/_cpso.py:405-426: in which generations it fires, nw, and exactly which rows it re-seeds)."""
    arrays = long_arrays()
    tag = case["tag"]
    w = arrays[tag + "__w"]
    looks, proj, rows8, count = set(case["looks"]), [], {}, [0]
    trace = []

    def cb(X, r):
        count[0] += 1
        trace.append(float(r.fun))
        if count[0] in looks:
            proj.append(np.asarray(X) @ w)
            rows8[str(count[0])] = X[:4, :8].copy()

    res = oracle.minimize(case["objective"], case_bounds(case), method=case["method"], options=dict(case["options"]),
                          callback=cb)
    restarts = list(zip([it for it, _ in res.get("_restarts", [])], res.get("_restart_rows", [])))
    ref = case["result"]
    assert np.array_equal(unhex(case["fun_trace"]), np.array(trace))
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    assert float(res.fun).hex() == ref["fun"] and np.array_equal(unhex(ref["x"]), res.x)
    # (X @ w is a BLAS product: its rounding, ~1e-14 here, depends on the library's blocking; a population that differs in
    #  ONE decision differs by O(search range) in that row's entry)
    assert np.allclose(arrays[tag + "__proj"], np.array(proj), rtol=0, atol=1e-9)
    for g, rows in case["pop_rows"].items():
        assert np.array_equal(unhex(rows), rows8[g])
    assert [[it, len(r)] for it, r in restarts] == case["restarts"]
    for it, r in restarts:
        assert np.array_equal(arrays[tag + "__restart_%d" % it], r)


PENALIZE = load_golden("cmaes_penalize.json")["cases"]


@pytest.mark.parametrize("case", PENALIZE, ids=lambda c: c["tag"])
def test_cmaes_penalize_bit_exact(case):
    """constraints="Penalize" (cmaes/_constraints.py:4-82): the reference's own test rows
    (tests/test_optimize.py:14-15) and boxes whose optimum lies on / outside the boundary."""
    trace = []
    res = _run(case, trace)
    ref = case["result"]
    assert np.array_equal(unhex(case["fun_trace"]), np.array([t[0] for t in trace]))
    assert (res.nit, res.nfev, res.status, res.message) == (ref["nit"], ref["nfev"], ref["status"], ref["message"])
    assert np.array_equal(unhex(ref["x"]), res.x) and float(res.fun).hex() == ref["fun"]
    arrays = np.load(os.path.join(GOLDEN, "cmaes_penalize_xall.npz"))
    assert np.array_equal(arrays[case["tag"] + "__xall"], res.xall)
    assert np.array_equal(arrays[case["tag"] + "__funall"], res.funall)
    lo, hi = np.transpose(case_bounds(case))
    assert np.all(res.xall + 1.0e-15 >= lo) and np.all(res.xall - 1.0e-15 <= hi)  # tests/helpers.py:23-25
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)


VDCMA = load_golden("vdcma.json")["cases"]


@pytest.mark.parametrize("case", VDCMA, ids=lambda c: c["tag"])
def test_vdcma_bit_exact(case):
    """VD-CMA (vdcma/_vdcma.py:144-460): the reference's own test rows (tests/test_optimize.py:119-132), with
    and without Penalize, and mid-size problems in which v and d adapt (n > 5) and the run stops on ftol."""
    trace = []
    res = _run(case, trace)
    ref = case["result"]
    assert np.array_equal(unhex(case["fun_trace"]), np.array([t[0] for t in trace]))
    assert (res.nit, res.nfev, res.status, res.message) == (ref["nit"], ref["nfev"], ref["status"], ref["message"])
    assert np.array_equal(unhex(ref["x"]), res.x) and float(res.fun).hex() == ref["fun"]
    arrays = np.load(os.path.join(GOLDEN, "vdcma_xall.npz"))
    assert np.array_equal(arrays[case["tag"] + "__xall"], res.xall)
    assert np.array_equal(arrays[case["tag"] + "__funall"], res.funall)
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)


WIDE = load_golden("vdcma_wide.json")


@pytest.mark.parametrize("case", WIDE["cases"], ids=lambda c: c["tag"])
def test_wide_rows_bit_exact(case):
    """Rows of more than 4096 elements (the reference has no dimension limit, de/_de.py:208-218): VD-CMA at n = 8192
    (vdcma/_vdcma.py:144-458) and short DE / PSO / CPSO runs at n = 4097 ... 9000, captured from the reference."""
    trace = []
    res = _run(case, trace)
    ref = case["result"]
    assert np.array_equal(unhex(case["fun_trace"]), np.array([t[0] for t in trace]))
    assert (res.nit, res.nfev, res.status, res.message) == (ref["nit"], ref["nfev"], ref["status"], ref["message"])
    arrays = np.load(os.path.join(GOLDEN, "vdcma_wide.npz"))
    assert np.array_equal(arrays[case["tag"] + "__x"], res.x) and float(res.fun).hex() == ref["fun"]
    if case["tag"] + "__pop_last" in arrays:
        assert np.array_equal(arrays[case["tag"] + "__pop_last"], trace[-1][1])


def test_wide_objective_values_bit_exact():
    """numpy's pairwise sums over rows of 4097 ... 65536 elements (32 ... 512 leaves), from the reference's functions."""
    from oracle.objectives import OBJECTIVES

    arrays = np.load(os.path.join(GOLDEN, "vdcma_wide.npz"))
    for key, want in WIDE["objective_kat"].items():
        name, n = key.rsplit("_", 1)
        assert np.array_equal(OBJECTIVES[name](arrays["kat_X_" + n]), unhex(want)), key


IMMEDIATE = load_golden("immediate.json")["cases"]


@pytest.mark.parametrize("case", IMMEDIATE, ids=lambda c: c["tag"])
def test_immediate_updating_bit_exact(case):
    """updating="immediate" (de_async / pso_async / selection_async): the reference's own test rows
    (tests/test_optimize.py:27-117) and mid-size problems for every strategy and constraint; the sphere rows
    pin the status rule (only the LAST individual of a sweep can end the run)."""
    trace = []
    res = _run(case, trace)
    ref = case["result"]
    assert np.array_equal(unhex(case["fun_trace"]), np.array([t[0] for t in trace]))
    assert (res.nit, res.nfev, res.status, res.message) == (ref["nit"], ref["nfev"], ref["status"], ref["message"])
    assert np.array_equal(unhex(ref["x"]), res.x) and float(res.fun).hex() == ref["fun"]
    arrays = np.load(os.path.join(GOLDEN, "immediate_xall.npz"))
    assert np.array_equal(arrays[case["tag"] + "__xall"], res.xall)
    assert np.array_equal(arrays[case["tag"] + "__funall"], res.funall)
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)


def test_populations_bit_exact():
    arrays = np.load(os.path.join(GOLDEN, "configs_pops.npz"))
    by_tag = {c["tag"]: c for c in CONFIGS}
    for key in arrays.files:
        tag = key[: -len("__pops")]
        trace = []
        _run(by_tag[tag], trace)
        assert np.array_equal(arrays[key], np.array([t[1] for t in trace])), tag


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    from oracle.streams import philox4x32_10

    def run(c, k):
        return tuple(int(v) for v in philox4x32_10(*c, *k))

    assert run((0, 0, 0, 0), (0, 0)) == (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)
    assert run((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)
    assert run((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == (
        0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)


def test_philox_stream_properties():
    s = oracle.PhiloxStream(12345)
    d = s.de_generation(2, 64, 19, 5, None)
    don = d["donors"]
    rows = np.arange(64)
    assert don.min() >= 0 and don.max() < 64
    for t in range(5):
        assert not np.any(don[t] == rows)
        for u in range(t):
            assert not np.any(don[t] == don[u])
    assert d["r1"].min() >= 0.0 and d["r1"].max() < 1.0
    assert d["irand"].min() >= 0 and d["irand"].max() < 19
    z = s.cma_normals(1, 512, 64)
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1.0) < 0.02
    # sharding invariance: rows of a shard see the same uniforms as in the whole population
    a = s.pso_generation(3, 64, 19)[0]
    b = s.pso_generation(3, 16, 19, row0=32)[0]
    assert np.array_equal(a[32:48], b)


NA_CASES = load_golden("na.json")["cases"]


@pytest.mark.parametrize("case", NA_CASES, ids=lambda c: c["tag"])
def test_na_oracle_matches_reference_bit_for_bit(case):
    """Neighbourhood Algorithm (na/_na.py:131-305): result, per-generation best-f, the full history and the
    populations handed to the callback equal the reference's, bit for bit -- including the reference's scalar
    `** 2` (libm pow) in the cell walk, the normalised rows it stores at iteration 1, and fixed axes."""
    trace, pops = [], []
    res = oracle.minimize(case["objective"], case_bounds(case), x0=case["x0"], method="na", options=dict(case["options"]),
                          callback=lambda X, r: (trace.append(float(r.fun)), pops.append(np.array(X, copy=True))))
    ref = case["result"]
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    assert np.array_equal(res.x, unhex(ref["x"])) and res.fun == unhex(ref["fun"])
    assert np.array_equal(np.array(trace), unhex(case["fun_trace"]))
    arrays = np.load(os.path.join(GOLDEN, "na_xall.npz"))
    assert np.array_equal(res.xall, arrays[case["tag"] + "__xall"])
    assert np.array_equal(res.funall, arrays[case["tag"] + "__funall"])
    import hashlib

    assert hashlib.sha256(np.ascontiguousarray(pops[-1]).tobytes()).hexdigest() == case["pop_last_sha"]
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)


@pytest.mark.parametrize("method", ["de", "pso", "cpso"])
def test_default_updating_is_the_references(method):
    """oracle.minimize without `updating` runs the reference's default (de/_de.py:27, cpso/_cpso.py:29: "immediate"), not the
    synchronous form the throughput paths use -- a caller comparing the two packages' defaults compares the same algorithm."""
    opts = {"maxiter": 12, "popsize": 10, "seed": 3}
    bounds = [[-5.12, 5.12]] * 6
    default = oracle.minimize("ackley", bounds, method=method, options=dict(opts))
    immediate = oracle.minimize("ackley", bounds, method=method, options=dict(opts, updating="immediate"))
    deferred = oracle.minimize("ackley", bounds, method=method, options=dict(opts, updating="deferred"))
    assert default["fun"] == immediate["fun"] and np.array_equal(default["x"], immediate["x"])
    assert default["fun"] != deferred["fun"]
