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


def test_vdcma_large_dimension_runs(sa):
    """The O(n) model at a dimension where full CMA-ES would need a 4096 x 4096 eigendecomposition per generation."""
    n = 4096
    res = sa.optimize.minimize(sa.factory.sphere, [[-5.12, 5.12]] * n, method="vdcma",
                               options={"maxiter": 30, "popsize": 64, "seed": 2, "rng": "philox", "sigma": 0.3})
    assert res.nit == 30 and np.isfinite(res.fun) and res.x.shape == (n,)
