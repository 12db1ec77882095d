"""GPU parity: VD-CMA (device sampling + objective, host model update) vs the reference's golden vectors and
the oracle (Philox draws)."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, case_bounds, load_golden, unhex

pytestmark = pytest.mark.gpu

CASES = load_golden("vdcma.json")["cases"]


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


def test_vd_sample_kernel_vs_numpy(sa):
    """sx_vdcma_sample against the numpy expression of vdcma/_vdcma.py:237-248, with and without injection."""
    import ctypes as C

    from stochopy_amd import _device

    ctx = _device.Context()
    rs = np.random.RandomState(3)
    for P, n, inject, row0 in ((7, 5, False, 0), (33, 130, True, 0), (12, 64, True, 6), (300, 1000, False, 0)):
        Z, d, v = rs.randn(P, n), rs.uniform(0.5, 2.0, n), rs.randn(n) / np.sqrt(n)
        xmean, sigma, dy = rs.uniform(-1, 1, n), 0.37, rs.randn(n)
        nv2 = v @ v
        vn = v / np.sqrt(nv2)
        coef = np.sqrt(1.0 + nv2) - 1.0
        want_y = d * (Z + coef * np.outer(Z @ vn, vn))
        if inject:
            for g, s in ((0, 1.0), (1, -1.0)):
                if 0 <= g - row0 < P:
                    want_y[g - row0] = s * dy
        bufs = [ctx.upload(a) for a in (Z, d, vn, xmean, dy)]
        ary, arx = ctx.empty((P, n)), ctx.empty((P, n))
        p = _device.ptr
        rc = ctx.L.sx_vdcma_sample(p(bufs[0]), P, n, row0, p(bufs[1]), p(bufs[2]), float(coef), p(bufs[3]), sigma,
                                   p(bufs[4]) if inject else None, p(ary), p(arx), ctx.stream_ptr)
        assert rc == 0
        ctx.sync()
        assert np.allclose(ary.cpu().numpy(), want_y, rtol=1e-13, atol=1e-14)
        assert np.allclose(arx.cpu().numpy(), xmean + sigma * want_y, rtol=1e-13, atol=1e-14)


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["tag"])
def test_vdcma_matches_reference_golden(sa, case):
    """numpy-legacy stream: the reference's per-generation best-f, history and result within the north-star
    tolerance (1e-6 rel; the only differences are summation orders of dot products)."""
    trace = []
    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), case_bounds(case), x0=case["x0"], method="vdcma",
                               options=opts, callback=lambda X, r: trace.append(float(r.fun)))
    ref = case["result"]
    want = unhex(case["fun_trace"])
    assert len(trace) == len(want) and np.allclose(trace, want, rtol=1e-6, atol=1e-300)
    assert (res.nit, res.nfev, res.status, res.message) == (ref["nit"], ref["nfev"], ref["status"], ref["message"])
    assert np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=0)
    assert np.allclose(res.x, unhex(ref["x"]), rtol=1e-5, atol=1e-7)
    arrays = np.load(os.path.join(GOLDEN, "vdcma_xall.npz"))
    assert np.allclose(res.funall, arrays[case["tag"] + "__funall"], rtol=1e-6, atol=1e-300)
    assert np.allclose(res.xall, arrays[case["tag"] + "__xall"], rtol=1e-5, atol=1e-6)
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)
    if case["options"].get("constraints") == "Penalize":
        lo, hi = np.transpose(case_bounds(case))
        assert np.all(res.xall + 1.0e-15 >= lo) and np.all(res.xall - 1.0e-15 <= hi)


@pytest.mark.parametrize("constraints", [None, "Penalize"])
def test_vdcma_philox_vs_oracle(sa, constraints):
    n, P = 30, 20
    bounds = [[-2.0, 3.0]] * n if constraints is None else [[0.5, 3.0]] * n
    opts = {"maxiter": 50, "popsize": P, "seed": 77, "sigma": 0.25, "constraints": constraints}
    t_ref, t_got = [], []
    ref = oracle.minimize("rosenbrock", bounds, method="vdcma", options=dict(opts), rng="philox",
                          callback=lambda X, r: t_ref.append(r.fun))
    got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="vdcma",
                               options=dict(opts, backend="hip", rng="philox"), callback=lambda X, r: t_got.append(r.fun))
    assert np.allclose(t_got, t_ref, rtol=1e-6) and (got.nit, got.status) == (ref.nit, ref.status)
    assert np.allclose(got.x, ref.x, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("obj,n,P,lo,hi,maxiter,verbosity", [("rosenbrock", 30, 20, 0.5, 3.0, 50, 1.0), ("sphere", 12, 10, 1.0, 4.0, 80, 0.0),
                                                             ("rastrigin", 200, 24, 0.2, 5.12, 30, 0.5)])
def test_vdcma_penalize_in_the_device_resident_loop_vs_oracle(sa, obj, n, P, lo, hi, maxiter, verbosity, monkeypatch):
    """constraints="Penalize" without a callback stays in the device-resident VD-CMA loop since round 3 (the boundary-weight
    bookkeeping is CMA-ES's cma_penalty_kernel with the diagonal of D (I + v v^T) D): stopping generation, status, the
    history of clipped points and penalised fitness, and the result against the oracle."""
    from stochopy_amd.optimize import _vdcma

    taken = []
    orig = _vdcma._VdDeviceRun.__init__

    def spy(self, *a, **k):
        taken.append(k.get("penalize"))
        orig(self, *a, **k)

    monkeypatch.setattr(_vdcma._VdDeviceRun, "__init__", spy)
    bounds = [[lo, hi]] * n
    opts = {"maxiter": maxiter, "popsize": P, "seed": 77, "sigma": 0.25, "constraints": "Penalize", "return_all": True,
            "verbosity": verbosity, "ftol": -1.0, "xtol": 0.0}
    ref = oracle.minimize(obj, bounds, method="vdcma", options=dict(opts), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, obj), bounds, method="vdcma", options=dict(opts, backend="hip", rng="philox"))
    assert taken == [True]
    assert (got.nit, got.nfev, got.status) == (ref.nit, ref.nfev, ref.status)
    assert got.funall.shape == ref.funall.shape
    assert np.allclose(got.funall, ref.funall, rtol=1e-6, atol=1e-300), np.abs(got.funall / ref.funall - 1).max()
    assert np.allclose(got.xall, ref.xall, rtol=1e-5, atol=1e-6 * (hi - lo))
    assert np.isclose(got.fun, ref.fun, rtol=1e-6) and np.allclose(got.x, ref.x, rtol=1e-5, atol=1e-6 * (hi - lo))
    assert np.all(got.xall >= lo - 1e-15) and np.all(got.xall <= hi + 1e-15)


def test_vdcma_large_dimension_runs(sa):
    """The O(n) model at a dimension where full CMA-ES would need a 4096 x 4096 eigendecomposition per generation."""
    n = 4096
    res = sa.optimize.minimize(sa.factory.sphere, [[-5.12, 5.12]] * n, method="vdcma",
                               options={"maxiter": 30, "popsize": 64, "seed": 2, "rng": "philox", "sigma": 0.3})
    assert res.nit == 30 and np.isfinite(res.fun) and res.x.shape == (n,)


@pytest.mark.parametrize("objective,n,P,maxiter", [("rosenbrock", 12, 16, 80), ("sphere", 40, 10, 60), ("rastrigin", 130, 24, 50),
                                                   ("rosenbrock", 600, 32, 30), ("sphere", 8, 6, 120)])
def test_device_loop_generation_by_generation_from_the_oracles_state(sa, objective, n, P, maxiter):
    """The device-resident VD-CMA generation (sx_vdcma_generation) checked one generation at a time: every generation
    starts from the ORACLE's model of that generation (mean, step size, rank-gap path, last mean shift, d, v, pc,
    best-f history, injection flag) and must arrive at the oracle's next model -- candidates, steps, fitness, best row,
    mean, shift, sigma, ps, pc, d, v, status -- to rounding."""
    import torch

    from stochopy_amd import _lib
    from stochopy_amd.optimize._vdcma import _VdDeviceRun

    seed, sigma0 = 99, 0.3
    bounds = np.array([[-3.0, 4.0]] * n)
    steps = []
    oracle.minimize(objective, bounds, method="vdcma", rng="philox",
                    options=dict(maxiter=maxiter, popsize=P, sigma=sigma0, seed=seed, xtol=1e-12, ftol=1e-30,
                                 probe=lambda it, before, after: steps.append((it, before, after))))
    assert len(steps) >= min(maxiter, 25)
    run = _VdDeviceRun(getattr(sa.factory, objective).sx_id, bounds[:, 0].copy(), bounds[:, 1].copy(), None, maxiter, P,
                       sigma0, 0.5, 1e-12, 1e-30, seed, run=False)
    buf = run.buffers

    def put(name, value):
        buf[name].copy_(torch.from_numpy(np.array(value, dtype=np.float64, order="C", copy=True)))

    def close(a, b, tol):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)

    with torch.cuda.stream(run.ctx.stream):
        for it, before, after in steps:
            for name in ("xmean", "dx", "dvec", "vvec", "vn", "pc", "besthist"):
                put(name, before[name])
            st = _lib.SxCmaState(it=it - 1, nfev=(it - 1) * P, best_row=0, fbest=0.0, sigma=before["sigma"],
                                 sigma_next=before["sigma"], tmp_coef=0.0, psnorm=0.0, status=_lib.SX_STATUS_NONE, done=0,
                                 stop_it=0)
            st.reserved[0], st.reserved[1], st.reserved[2] = before["ps"], before["norm_v2"], before["norm_v"]
            st.reserved[3], st.reserved[4] = float(before["inject"]), float(np.sqrt(1.0 + before["norm_v2"]) - 1.0)
            put("state", np.frombuffer(bytes(st), dtype=np.float64))
            run.step(it)
            got = run.read_state()
            best = int(after["order"][0])
            assert close(buf["ary"].cpu().numpy(), after["ary"], 1e-11), it
            assert close(buf["arx"].cpu().numpy(), after["arx"], 1e-11), it
            assert np.allclose(buf["fit"].cpu().numpy(), after["arfit"], rtol=1e-9, atol=1e-300), it
            assert np.array_equal(buf["order"].cpu().numpy(), after["order"]), it
            assert (got.it, got.nfev, got.best_row) == (it, it * P, best), it
            assert np.isclose(got.fbest, after["arfit"][best], rtol=1e-9, atol=0), it
            for name, tol in (("xmean", 1e-11), ("dx", 1e-9), ("pc", 1e-9), ("dvec", 1e-9), ("vvec", 1e-9), ("vn", 1e-9)):
                assert close(buf[name].cpu().numpy(), after[name], tol), (it, name)
            assert np.isclose(got.sigma, after["sigma"], rtol=1e-10, atol=0), it
            assert np.isclose(got.reserved[0], after["ps"], rtol=1e-10, atol=1e-300), it
            assert np.isclose(got.reserved[1], after["norm_v2"], rtol=1e-9, atol=0), it
            assert got.reserved[3] == 1.0
            want = _lib.SX_STATUS_NONE if after["status"] is None else after["status"]
            assert (got.status, bool(got.done)) == (want, after["status"] is not None), it


@pytest.mark.parametrize("objective,n,P,maxiter,extra", [("rosenbrock", 12, 16, 150, {}), ("sphere", 40, 10, 400, {"ftol": 1e-9}),
                                                         ("rastrigin", 130, 24, 60, {"return_all": True}),
                                                         ("rosenbrock", 700, 64, 40, {"return_all": True, "verbosity": 0.0}),
                                                         ("sphere", 2000, 16, 25, {})])
def test_vdcma_device_loop_matches_oracle(sa, objective, n, P, maxiter, extra, monkeypatch):
    """Whole runs through the device-resident loop (Philox draws, no callback) against the oracle: same stopping
    generation and status, best-f / result / histories within the north-star tolerance (1e-6 rel)."""
    opts = dict(maxiter=maxiter, popsize=P, sigma=0.3, seed=7, **extra)
    bounds = [[-3.0, 4.0]] * n
    ref = oracle.minimize(objective, bounds, method="vdcma", options=dict(opts), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="vdcma",
                               options=dict(opts, backend="hip", rng="philox"))
    assert (got.nit, got.nfev, got.status) == (ref["nit"], ref["nfev"], ref["status"])
    assert np.isclose(got.fun, ref["fun"], rtol=1e-6, atol=1e-300)
    assert np.allclose(got.x, ref["x"], rtol=1e-5, atol=1e-7)
    if extra.get("return_all"):
        assert got.xall.shape == ref["xall"].shape
        assert np.allclose(got.funall, ref["funall"], rtol=1e-6, atol=1e-300)
        assert np.allclose(got.xall, ref["xall"], rtol=1e-5, atol=1e-6)
    # a callback is served from the same loop (round 3): every generation's candidates and best, as the oracle's callback sees them
    seen_ref, seen_got = [], []
    oracle.minimize(objective, bounds, method="vdcma", options=dict(opts), rng="philox",
                    callback=lambda X, r: seen_ref.append((np.array(X), np.array(r.x), float(r.fun), int(r.nfev), int(r.nit))))
    cb = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="vdcma", options=dict(opts, backend="hip", rng="philox"),
                              callback=lambda X, r: seen_got.append((np.array(X), np.array(r.x), float(r.fun), int(r.nfev), int(r.nit))))
    assert (cb.nit, cb.status) == (got.nit, got.status) and cb.fun == got.fun and np.array_equal(cb.x, got.x)
    assert len(seen_got) == len(seen_ref) == got.nit
    for (X, x, f, nfev, nit), (Xr, xr, fr, nfevr, nitr) in zip(seen_got, seen_ref):
        assert (nfev, nit) == (nfevr, nitr) and X.shape == Xr.shape == (P, n)
        assert np.isclose(f, fr, rtol=1e-6, atol=1e-300) and np.allclose(X, Xr, rtol=1e-5, atol=1e-6) and np.allclose(x, xr, rtol=1e-5, atol=1e-6)
    # and the host-driven loop (SX_CMA_LOOP=host; legacy draws take it too) gives the same run
    monkeypatch.setenv("SX_CMA_LOOP", "host")
    trace = []
    hl = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="vdcma",
                              options=dict(opts, backend="hip", rng="philox"), callback=lambda X, r: trace.append(r.fun))
    assert (hl.nit, hl.status) == (got.nit, got.status) and np.isclose(hl.fun, got.fun, rtol=1e-6, atol=1e-300)


def test_vdcma_device_loop_small_shapes_and_short_runs(sa):
    """n = 2 ... 7 (n <= 5: the learning rates c1, cmu are zero or negative, so d and v stay put), popsize 2 ... 8,
    1 ... 30 generations: the device-resident loop stops where the oracle stops, with its result."""
    for n in (2, 3, 5, 6, 7):
        for P in (2, 3, 8):
            for maxiter in (1, 2, 5, 30):
                o = dict(maxiter=maxiter, popsize=P, sigma=0.3, seed=3)
                b = [[-2.0, 3.0]] * n
                ref = oracle.minimize("sphere", b, method="vdcma", options=dict(o), rng="philox")
                got = sa.optimize.minimize(sa.factory.sphere, b, method="vdcma", options=dict(o, backend="hip", rng="philox"))
                assert (got.nit, got.status) == (ref.nit, ref.status), (n, P, maxiter)
                assert np.isclose(got.fun, ref.fun, rtol=1e-6, atol=1e-12), (n, P, maxiter)
