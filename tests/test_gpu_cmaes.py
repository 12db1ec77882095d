"""GPU parity: CMA-ES kernels (fp64 MFMA sampling / covariance update) and the CMA-ES loop."""
import numpy as np
import pytest

import oracle
from oracle import engine as oe
from conftest import case_bounds, load_golden, unhex

pytestmark = pytest.mark.gpu

# floating-point tolerance of the MFMA contractions vs numpy/BLAS: same products, different summation
# order => relative to the magnitude of the terms
GEMM_RTOL = 1e-13


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


@pytest.fixture(scope="module")
def ctx(sa):
    from stochopy_amd import _device

    return _device.Context()


# the last four reach the 32x64 and 64x64 workgroup tiles (whole and ragged), which the CMA-ES shapes below n = 1024 never launch
@pytest.mark.parametrize("shape", [(10, 2), (7, 3), (64, 64), (65, 33), (100, 130), (1024, 512), (300, 257), (1024, 1024), (1000, 1030),
                                   (2048, 1024), (2050, 1030)])
def test_sample_kernel_vs_numpy(sa, ctx, shape):
    from stochopy_amd import _device, _lib

    P, n = shape
    rs = np.random.RandomState(P + n)
    Z = rs.randn(P, n)
    Bm = np.linalg.qr(rs.randn(n, n))[0]
    D = rs.uniform(0.5, 2.0, n)
    xmean = rs.uniform(-1, 1, n)
    sigma = 0.37
    ref = oe.cma_sample(xmean, sigma, Bm, D, Z)
    d = {k: ctx.upload(v) for k, v in dict(Z=Z, B=Bm, D=D, xm=xmean).items()}
    out = ctx.empty((P, n))
    p = _device.ptr
    _lib.check(ctx.L.sx_cmaes_sample(p(d["xm"]), sigma, p(d["B"]), p(d["D"]), p(d["Z"]), p(out), P, n, ctx.stream_ptr))
    ctx.sync()
    got = out.cpu().numpy()
    scale = np.abs(xmean).max() + sigma * ((np.abs(Z) * D) @ np.abs(Bm).T).max()
    assert np.abs(got - ref).max() <= GEMM_RTOL * scale


# (64, 1472, 64) and (70, 1475, 67): the 64x64 tile of the covariance update (n >= 1408)
@pytest.mark.parametrize("shape", [(10, 2, 5), (12, 3, 6), (64, 64, 32), (200, 130, 77), (1024, 512, 512), (90, 257, 45), (64, 1472, 64),
                                   (70, 1475, 67)])
@pytest.mark.parametrize("cond", [True, False])
def test_rank_mu_and_recombine_vs_numpy(sa, ctx, shape, cond):
    from stochopy_amd import _device, _lib

    P, n, mu = shape
    rs = np.random.RandomState(P * 3 + n)
    arx = rs.randn(P, n)
    order = rs.permutation(P)
    w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    w /= w.sum()
    xold = rs.randn(n) * 0.1
    pc = rs.randn(n) * 0.3
    A = rs.randn(n, n)
    C0 = A @ A.T / n + np.eye(n)
    sigma, c1, cmu, cc = 0.21, 0.013, 0.11, 0.2
    ref = oe.cma_covariance(C0.copy(), arx[order[:mu]], xold, sigma, w, pc, cond, c1, cmu, cc)
    ref_mean = np.dot(w, arx[order[:mu]])
    p = _device.ptr
    d_arx, d_w, d_xold, d_pc, d_C = (ctx.upload(v) for v in (arx, w, xold, pc, C0))
    d_idx = ctx.upload(np.ascontiguousarray(order[:mu], dtype=np.int64))
    tmp = 0.0 if cond else c1 * cc * (2.0 - cc)
    d_Y = ctx.empty((mu, n))
    _lib.check(ctx.L.sx_cmaes_rank_mu(p(d_arx), p(d_idx), p(d_w), mu, p(d_xold), sigma, p(d_pc), c1, cmu, tmp, p(d_C),
                                      p(d_Y), n, ctx.stream_ptr))
    d_mean = ctx.empty((n,))
    _lib.check(ctx.L.sx_cmaes_recombine(p(d_arx), p(d_idx), p(d_w), mu, n, p(d_mean), ctx.stream_ptr))
    ctx.sync()
    got = d_C.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.abs(d_mean.cpu().numpy() - ref_mean).max() <= 1e-13 * (np.abs(arx).max())
    # symmetrise: C = triu(C) + triu(C,1).T
    _lib.check(ctx.L.sx_symmetrize_upper(p(d_C), n, ctx.stream_ptr))
    ctx.sync()
    assert np.array_equal(d_C.cpu().numpy(), np.triu(got) + np.triu(got, 1).T)


def test_philox_normals_vs_oracle(sa, ctx):
    from stochopy_amd import _device, _lib

    P, n, seed, gen = 33, 200, 987654321012, 7
    ref = oracle.PhiloxStream(seed).cma_normals(gen, P, n)
    out = ctx.empty((P, n))
    _lib.check(ctx.L.sx_cmaes_normals(_device.ptr(out), P, n, 0, gen, seed & 0xFFFFFFFF, seed >> 32, ctx.stream_ptr))
    ctx.sync()
    assert np.allclose(out.cpu().numpy(), ref, rtol=1e-13, atol=1e-15)  # device log/sin/cos vs libm: a few ulp


CMA_CASES = [c for c in load_golden("configs.json")["cases"] if c["method"] == "cmaes"]


def _eigenbasis_is_determined(case):
    """While C = I + (rank mu+1 update) and mu + 1 < n, the eigenvalue 1-c1-cmu is repeated and LAPACK's
    basis of that eigenspace depends on rounding noise in C (summation order of the contraction): the
    realised samples B*(D o z) are then not reproducible across BLAS builds either.  Same-seed parity is
    only meaningful when mu + 1 >= n (every BASELINE.json config, every reference-suite xref)."""
    P = case["options"]["popsize"]
    mu = int(case["options"].get("muperc", 0.5) * P)
    return mu + 1 >= case["ndim"]


@pytest.mark.parametrize("case", CMA_CASES, ids=lambda c: c["tag"])
def test_cmaes_matches_reference_golden(sa, case):
    """numpy-legacy stream + host LAPACK eigh: the reference's per-generation best-f within 1e-6 rel
    (north-star tolerance; the contractions differ from BLAS only in summation order)."""
    trace = []
    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), case_bounds(case), x0=case["x0"], method="cmaes",
                               options=opts, callback=lambda X, r: trace.append(float(r.fun)))
    ref = case["result"]
    want = unhex(case["fun_trace"])
    if _eigenbasis_is_determined(case):
        assert len(trace) == len(want)
        assert np.allclose(trace, want, rtol=1e-6, atol=1e-300)
        assert (res.nit, res.nfev, res.status, res.success, res.message) == (
            ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
        assert np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=0)
    else:
        # NOT a parity check, a property test: generation 1 (B = I) is exact; afterwards only the distribution is determined
        # (same-seed parity for these two cases: test_cmaes_golden_with_the_references_eigenpairs_replayed below)
        assert np.isclose(trace[0], want[0], rtol=1e-12)
        assert res.status == ref["status"]
        assert res.fun <= max(10.0 * unhex(ref["fun"]), case["options"].get("ftol", 1e-8))


def test_cmaes_c4_long_run_follows_the_reference(sa):
    """BASELINE config 4 (n=512, P=1024; mu + 1 >= n: the eigenbasis is determined) over 16 generations against the
    reference's vectors: best-f of every generation, the final x, and ALL candidates of every generation (projection)."""
    from conftest import check_long_case

    case = {c["tag"]: c for c in load_golden("configs_long.json")["cases"]}["C4L_cmaes_rosen_n512_p1024"]
    check_long_case(sa, case)


@pytest.mark.parametrize("case", CMA_CASES, ids=lambda c: c["tag"])
def test_cmaes_golden_with_the_references_eigenpairs_replayed(sa, case):
    """Same-seed parity for EVERY CMA-ES golden, the two with a repeated eigenvalue (mu + 1 < n) included: the oracle
    -- bit-identical to the reference on these cases (tests/test_oracle_golden.py) -- records the (C, eigenvalues,
    eigenvectors) of each of its LAPACK calls, and the HIP run gets those pairs back through ``eigh=callable``
    instead of decomposing its own C.  What is left to differ is everything the GPU computes -- normals upload,
    sampling GEMM, objective, recombination, rank-mu update (checked here: its C against the oracle's at 1e-9) -- and
    the run must follow the reference's per-generation best-f within 1e-6 to the last generation."""
    recorded = []

    def record(Cmat):
        w, V = np.linalg.eigh(Cmat)
        recorded.append((Cmat.copy(), w, V))
        return w, V

    oracle.minimize(case["objective"], case_bounds(case), x0=case["x0"], method="cmaes",
                    options=dict(case["options"], eigh=record), rng="numpy-legacy")
    calls = []

    def replay(Cmat):
        Cref, w, V = recorded[len(calls)]
        calls.append(float(np.abs(Cmat - Cref).max() / np.abs(Cref).max()))
        return w, V

    trace = []
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), case_bounds(case), x0=case["x0"], method="cmaes",
                               options=dict(case["options"], backend="hip", rng="numpy-legacy", eigh=replay),
                               callback=lambda X, r: trace.append(float(r.fun)))
    ref, want = case["result"], unhex(case["fun_trace"])
    assert len(calls) == len(recorded) and max(calls) <= 1e-9
    assert len(trace) == len(want) and np.allclose(trace, want, rtol=1e-6, atol=1e-300)
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    assert np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=0)
    xref = np.asarray(unhex(ref["x"]))  # (the C4 fixture keeps the leading coordinates only)
    assert np.allclose(res.x[: xref.size], xref, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tag", ["cmaes_none", "cmaes_none_x0"])
def test_cmaes_reference_suite_xrefs(sa, tag):
    """reference tests/test_optimize.py:9-20 (constraints=None rows)."""
    case = {c["tag"]: c for c in load_golden("suite_rosen2d.json")["cases"]}[tag]
    opts = dict(case["options"], backend="hip")
    res = sa.optimize.minimize(sa.factory.rosenbrock, case_bounds(case), x0=case["x0"], method="cmaes", options=opts)
    assert np.allclose(case["xref_from_reference_tests"], res.x)
    assert res.xall.shape[1:] == (8, 2)


PENALIZE_CASES = load_golden("cmaes_penalize.json")["cases"]


@pytest.mark.parametrize("case", PENALIZE_CASES, ids=lambda c: c["tag"])
def test_cmaes_penalize_matches_reference_golden(sa, case):
    """constraints="Penalize": clipping + objective + weighted squared excess on the device, boundary-weight
    bookkeeping on the host.  The reference's per-generation best-f, its full history and its result within the
    north-star tolerance (1e-6 rel); every visible point inside the box (reference tests/helpers.py:23-25)."""
    import os

    from conftest import GOLDEN

    assert _eigenbasis_is_determined(case)
    trace = []
    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), case_bounds(case), x0=case["x0"], method="cmaes",
                               options=opts, callback=lambda X, r: trace.append(float(r.fun)))
    ref = case["result"]
    want = unhex(case["fun_trace"])
    assert len(trace) == len(want) and np.allclose(trace, want, rtol=1e-6, atol=1e-300)
    assert (res.nit, res.nfev, res.status, res.message) == (ref["nit"], ref["nfev"], ref["status"], ref["message"])
    assert np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=0)
    assert np.allclose(res.x, unhex(ref["x"]), rtol=1e-5, atol=1e-7)
    arrays = np.load(os.path.join(GOLDEN, "cmaes_penalize_xall.npz"))
    assert np.allclose(res.funall, arrays[case["tag"] + "__funall"], rtol=1e-6, atol=1e-300)
    assert np.allclose(res.xall, arrays[case["tag"] + "__xall"], rtol=1e-5, atol=1e-6)
    lo, hi = np.transpose(case_bounds(case))
    assert np.all(res.xall + 1.0e-15 >= lo) and np.all(res.xall - 1.0e-15 <= hi)
    if "xref_from_reference_tests" in case:
        assert np.allclose(case["xref_from_reference_tests"], res.x)


def test_cmaes_penalize_philox_and_sharded_eval_shapes(sa):
    """Philox draws: hip == oracle within tolerance while the mean sits outside the box (weights active)."""
    n, P = 6, 10
    bounds = [[1.0, 5.0]] * n
    opts = {"maxiter": 60, "popsize": P, "seed": 99, "sigma": 0.3, "constraints": "Penalize"}
    t_ref, t_got = [], []
    ref = oracle.minimize("sphere", bounds, method="cmaes", options=dict(opts), rng="philox",
                          callback=lambda X, r: t_ref.append(r.fun))
    got = sa.optimize.minimize(sa.factory.sphere, bounds, method="cmaes",
                               options=dict(opts, backend="hip", rng="philox", eigh="host"),  # the oracle's eigenbasis
                               callback=lambda X, r: t_got.append(r.fun))
    assert np.allclose(t_got, t_ref, rtol=1e-6) and got.nit == ref.nit and got.status == ref.status
    assert np.all(got.x >= 1.0 - 1e-15) and np.allclose(got.x, ref.x, rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("obj,n,P,lo,hi,maxiter,verbosity", [
    ("sphere", 6, 12, 1.0, 5.0, 80, 1.0),       # the optimum sits on the boundary: the mean leaves the box, weights grow
    ("rosenbrock", 10, 24, -0.5, 0.8, 60, 0.5),  # optimum (1, .., 1) outside: active penalties throughout
    ("sphere", 40, 82, 2.0, 3.0, 30, 0.0),       # blocks of the eigensolver's small path, mu + 1 >= n
    ("rastrigin", 130, 264, 0.3, 5.12, 16, 1.0),  # the block eigensolver (n > 64)
])
def test_cmaes_penalize_in_the_device_resident_loop_vs_oracle(sa, obj, n, P, lo, hi, maxiter, verbosity, monkeypatch):
    """constraints="Penalize" with Philox draws and no callback stays on the device since round 3:
    clipped objective, percentiles of the raw fitness, spread history + median, boundary weights, weighted squared excess
    (csrc/sx_cma_loop.hip cma_penalty_kernel; cmaes/_constraints.py:4-82).  Against the oracle (LAPACK + canonical
    signs): stopping generation, status, the full history (clipped points and PENALISED fitness, as the reference
    stores them) and the result within the north-star tolerance; every visible point inside the box."""
    from stochopy_amd.optimize import _cmaes

    taken = []
    orig = _cmaes._CmaDeviceRun.__init__

    def spy(self, *a, **k):
        taken.append(k.get("penalize"))
        orig(self, *a, **k)

    monkeypatch.setattr(_cmaes._CmaDeviceRun, "__init__", spy)
    bounds = [[lo, hi]] * n
    opts = {"maxiter": maxiter, "popsize": P, "seed": 2024 + n, "sigma": 0.3, "constraints": "Penalize", "return_all": True,
            "verbosity": verbosity, "ftol": -1.0, "xtol": 0.0}
    ref = oracle.minimize(obj, bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, obj), bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
    assert taken == [True]  # the device-resident loop ran, with the penalty bookkeeping on the device
    assert (got.nit, got.nfev, got.status) == (ref.nit, ref.nfev, ref.status)
    assert got.funall.shape == ref.funall.shape and got.xall.shape == ref.xall.shape
    assert np.allclose(got.funall, ref.funall, rtol=1e-6, atol=1e-300), np.abs(got.funall / ref.funall - 1).max()
    assert np.allclose(got.xall, ref.xall, rtol=1e-5, atol=1e-6 * (hi - lo))
    assert np.isclose(got.fun, ref.fun, rtol=1e-6, atol=0) and np.allclose(got.x, ref.x, rtol=1e-5, atol=1e-6 * (hi - lo))
    assert np.all(got.xall >= lo - 1e-15) and np.all(got.xall <= hi + 1e-15) and np.all(got.x >= lo - 1e-15)


def test_cmaes_philox_vs_oracle(sa):
    n, P = 20, 48  # mu + 1 >= n: the eigenbasis is determined (see _eigenbasis_is_determined)
    opts = {"maxiter": 12, "popsize": P, "seed": 4242, "sigma": 0.2}
    bounds = [[-3.0, 3.0]] * n
    t_ref, t_got = [], []
    oracle.minimize("rosenbrock", bounds, method="cmaes", options=dict(opts), rng="philox",
                    callback=lambda X, r: t_ref.append(r.fun))
    sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="cmaes",
                         options=dict(opts, backend="hip", rng="philox", eigh="host"),  # the oracle's eigenbasis
                         callback=lambda X, r: t_got.append(r.fun))
    assert np.allclose(t_got, t_ref, rtol=1e-6)


@pytest.mark.parametrize("shape", [(2, 10), (5, 12), (20, 48), (33, 80), (70, 160), (130, 264)])
def test_cmaes_device_eigensolver_vs_oracle_canonical(sa, shape):
    """eigh="device" (csrc/sx_eigh.hip) against the oracle running the reference's LAPACK call with the same sign
    rule (oracle eigh="canonical"): same seed, per-generation best-f within 1e-6 rel.  mu + 1 >= n, so the
    eigenbasis is determined (see _eigenbasis_is_determined)."""
    n, P = shape
    opts = {"maxiter": 14, "popsize": P, "seed": 77 + n, "sigma": 0.2}
    bounds = [[-3.0, 3.0]] * n
    t_ref, t_got = [], []
    ref = oracle.minimize("rosenbrock", bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox",
                          callback=lambda X, r: t_ref.append(r.fun))
    got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="cmaes",
                               options=dict(opts, backend="hip", rng="philox", eigh="device"),
                               callback=lambda X, r: t_got.append(r.fun))
    assert len(t_got) == len(t_ref) and np.allclose(t_got, t_ref, rtol=1e-6)
    assert (got.nit, got.status) == (ref.nit, ref.status) and np.isclose(got.fun, ref.fun, rtol=1e-6)


def test_cmaes_c4_device_eigensolver_vs_oracle_canonical(sa):
    """BASELINE config 4 (CMA-ES Rosenbrock n=512 P=1024 seed 0, numpy-legacy draws) with the device eigensolver:
    the oracle with LAPACK + the same sign rule gives the same per-generation best-f within 1e-6 rel; generation 1
    (B = I) also equals the reference's golden value."""
    case = {c["tag"]: c for c in CMA_CASES}
    c4 = [c for c in CMA_CASES if c["ndim"] == 512][0]
    opts = dict(c4["options"])
    t_ref, t_got = [], []
    oracle.minimize(c4["objective"], case_bounds(c4), x0=c4["x0"], method="cmaes", options=dict(opts, eigh="canonical"),
                    callback=lambda X, r: t_ref.append(r.fun))
    sa.optimize.minimize(getattr(sa.factory, c4["objective"]), case_bounds(c4), x0=c4["x0"], method="cmaes",
                         options=dict(opts, backend="hip", rng="numpy-legacy", eigh="device"),
                         callback=lambda X, r: t_got.append(float(r.fun)))
    assert len(t_got) == len(t_ref) and np.allclose(t_got, t_ref, rtol=1e-6)
    assert np.isclose(t_got[0], unhex(c4["fun_trace"])[0], rtol=1e-12)
    assert len(case) == len(CMA_CASES)


@pytest.mark.parametrize("cfg", [("rosenbrock", 2, 10, 100, 0.1), ("rosenbrock", 20, 48, 60, 0.2), ("sphere", 6, 12, 400, 0.3),
                                 ("rastrigin", 33, 80, 40, 0.3), ("ackley", 70, 160, 30, 0.2), ("rosenbrock", 130, 264, 12, 0.2),
                                 # popsize / n one, two, three past a multiple of 64: the ragged last step of the voting rank kernels
                                 ("sphere", 10, 65, 30, 0.3), ("rosenbrock", 12, 130, 30, 0.2), ("sphere", 67, 195, 14, 0.3),
                                 ("rosenbrock", 129, 258, 10, 0.2)],
                         ids=lambda c: "%s_n%d_p%d" % c[:3])
def test_cmaes_device_resident_loop_vs_oracle(sa, cfg, monkeypatch):
    """No callback, no history, Philox draws: the whole loop runs on the device (csrc/sx_cma_loop.hip: ranking,
    recombination, paths, step size, stop rules; the host looks at the state every few generations).  Same seed as the
    oracle with LAPACK + the canonical sign rule: same stopping generation and status, best-f / best-x within the
    north-star tolerance."""
    obj, n, P, maxiter, sigma = cfg
    opts = {"maxiter": maxiter, "popsize": P, "seed": 1234 + n, "sigma": sigma}
    bounds = [[-3.0, 3.0]] * n
    ref = oracle.minimize(obj, bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, obj), bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
    assert (got.nit, got.nfev, got.status, got.success, got.message) == (ref.nit, ref.nfev, ref.status, ref.success, ref.message)
    assert np.isclose(got.fun, ref.fun, rtol=1e-6, atol=1e-300)
    assert np.allclose(got.x, ref.x, rtol=1e-5, atol=1e-7)
    # and the host-driven loop (legacy draws, eigh="host" / callable, or SX_CMA_LOOP=host) agrees with it
    trace = []
    monkeypatch.setenv("SX_CMA_LOOP", "host")
    via_host = sa.optimize.minimize(getattr(sa.factory, obj), bounds, method="cmaes",
                                    options=dict(opts, backend="hip", rng="philox"), callback=lambda X, r: trace.append(r.fun))
    assert (via_host.nit, via_host.status) == (got.nit, got.status) and np.isclose(via_host.fun, got.fun, rtol=1e-6)
    assert len(trace) == got.nit


@pytest.mark.parametrize("obj,n,P,maxiter,constraints,verbosity", [
    ("rosenbrock", 20, 48, 40, None, 1.0), ("sphere", 6, 12, 80, "Penalize", 0.5), ("rastrigin", 70, 160, 12, None, 0.0),
    ("rosenbrock", 130, 264, 10, "Penalize", 1.0)])
def test_cmaes_callback_on_the_device_resident_loop(sa, obj, n, P, maxiter, constraints, verbosity, monkeypatch):
    """A callback no longer sends a Philox / device-eigensolver run to the host-driven loop (VERDICT r2 missing #3): the
    generation stays on the device, the host looks at every generation and hands the callback what the reference does
    (cmaes/_cmaes.py:333-343): all candidates of the generation (the clipped ones with Penalize), un-standardised, and
    res.x / fun / nfev / nit of its best, with return_all the history so far.  Against the oracle's callback, generation
    by generation, and against the same run without a callback."""
    from stochopy_amd.optimize import _cmaes

    taken = []
    orig = _cmaes._CmaDeviceRun.__init__

    def spy(self, *a, **k):
        taken.append(k.get("callback") is not None)
        orig(self, *a, **k)

    monkeypatch.setattr(_cmaes._CmaDeviceRun, "__init__", spy)
    lo, hi = (1.0, 5.0) if constraints else (-3.0, 3.0)
    bounds = [[lo, hi]] * n
    opts = {"maxiter": maxiter, "popsize": P, "seed": 77 + n, "sigma": 0.3, "constraints": constraints, "return_all": True,
            "verbosity": verbosity, "ftol": -1.0, "xtol": 0.0}
    seen_ref, seen_got = [], []

    def keep(store, history):
        def cb(X, r):
            store.append((np.array(X), np.array(r.x), float(r.fun), int(r.nfev), int(r.nit),
                          (np.array(r.xall), np.array(r.funall)) if history else None))
        return cb

    # (the oracle's callback result carries no history; the reference's does: checked against the final arrays)
    ref = oracle.minimize(obj, bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox", callback=keep(seen_ref, False))
    got = sa.optimize.minimize(getattr(sa.factory, obj), bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"),
                               callback=keep(seen_got, True))
    assert taken == [True]
    assert (got.nit, got.nfev, got.status) == (ref.nit, ref.nfev, ref.status) and len(seen_got) == len(seen_ref) == got.nit
    for (X, x, f, nfev, nit, hist), (Xr, xr, fr, nfevr, nitr, _) in zip(seen_got, seen_ref):
        assert (nfev, nit) == (nfevr, nitr) and X.shape == Xr.shape == (P, n)
        assert np.isclose(f, fr, rtol=1e-6, atol=1e-300)
        assert np.allclose(X, Xr, rtol=1e-5, atol=1e-6 * (hi - lo)) and np.allclose(x, xr, rtol=1e-5, atol=1e-6 * (hi - lo))
        assert hist[0].shape == got.xall[:nit].shape and np.array_equal(hist[0], got.xall[:nit]) and np.array_equal(hist[1], got.funall[:nit])
        if constraints:
            assert X.min() >= lo - 1e-15 and X.max() <= hi + 1e-15
    assert np.allclose(got.funall, ref.funall, rtol=1e-6, atol=1e-300)
    plain = sa.optimize.minimize(getattr(sa.factory, obj), bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
    assert (plain.nit, plain.status) == (got.nit, got.status) and plain.fun == got.fun and np.array_equal(plain.x, got.x)
    assert np.array_equal(plain.xall, got.xall) and np.array_equal(plain.funall, got.funall)


@pytest.mark.parametrize("n,P,maxiter", [(512, 1024, 20), (257, 520, 24)], ids=["c4_n512_p1024", "n257_p520"])
def test_cmaes_device_resident_loop_block_eigensolver_vs_oracle(sa, n, P, maxiter):
    """BASELINE config 4 (Rosenbrock n=512, P=1024) and an odd size that pads (n=257) through the path whose
    throughput is quoted: device-resident loop, Philox draws, the BLOCK path of the eigensolver (n > 64:
    eigh_round_kernel, warm start from the previous basis through eigh_gemm_kernel, the measured stopping rule) in
    every generation.  Against the oracle with LAPACK + the canonical sign rule: same stopping generation and
    status, best-f of EVERY generation (device-side history, read back once) and the final best-x within the
    north-star tolerance.  mu >= n - 1 in both shapes, so the covariance has no exactly repeated eigenvalue and the
    basis is determined."""
    opts = {"maxiter": maxiter, "popsize": P, "seed": 0, "sigma": 0.1, "ftol": -1.0, "xtol": 0.0, "return_all": True,
            "verbosity": 0.0}
    bounds = [[-5.12, 5.12]] * n
    ref = oracle.minimize("rosenbrock", bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox")
    got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
    assert (got.nit, got.nfev, got.status) == (ref.nit, ref.nfev, ref.status) == (maxiter, maxiter * P, -1)
    assert got.funall.shape == ref.funall.shape
    assert np.allclose(got.funall, ref.funall, rtol=1e-6, atol=0.0), np.abs(got.funall / ref.funall - 1.0).max()
    assert np.isclose(got.fun, ref.fun, rtol=1e-6, atol=0.0)
    # best-x: 1e-6 of the search range per coordinate (the spectrum is clustered in these generations -- gaps of 1e-6
    # relative -- so eigenvector rounding differences of two solvers are amplified to ~1e-7 of the range by generation 20)
    assert np.abs(got.x - ref.x).max() <= 1e-6 * 10.24, np.abs(got.x - ref.x).max()


@pytest.mark.parametrize("objective,n,P,maxiter,ftol", [("rosenbrock", 300, 320, 40, -1.0), ("sphere", 512, 600, 30, -1.0),
                                                        ("sphere", 260, 64, 400, 1.0), ("sphere", 300, 64, 1, -1.0),
                                                        ("sphere", 300, 16, 25, -1.0), ("sphere", 300, 64, 30, 2500.0)])
def test_cmaes_decomposition_enqueued_in_pieces_is_the_same_run(sa, objective, n, P, maxiter, ftol, monkeypatch):
    """n > 256 on one GPU without a callback: every decomposition is enqueued in pieces (sx_cmaes_generation_phased: the
    rounds the last one needed, one look at the solver's run record, its finish or another sweep) and the state record is
    seen one generation late.  Same kernels on the same data minus the no-op launches: the run is, bit for bit, the one
    with whole decompositions enqueued ahead (SX_CMA_PHASED=0) -- incl. a stop by ftol in the middle of the run (the
    generation enqueued behind the stop does nothing) and the device-side history."""
    from stochopy_amd import _lib

    opts = {"maxiter": maxiter, "popsize": P, "seed": 2, "sigma": 0.1, "ftol": ftol, "xtol": 0.0, "return_all": True,
            "verbosity": 0.0, "backend": "hip", "rng": "philox"}
    bounds = [[-5.12, 5.12]] * n
    calls = []
    L = _lib.lib()
    orig = L.sx_cmaes_generation_phased

    class Spy:
        def __call__(self, *a):
            calls.append(int(a[3]))
            return orig(*a)

    monkeypatch.setattr(L, "sx_cmaes_generation_phased", Spy())
    pieces = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="cmaes", options=dict(opts))
    # (every due decomposition: one start, one finish -- the last start may belong to a generation behind the stop)
    assert calls.count(0) >= 1 and 0 <= calls.count(0) - calls.count(2) <= 1, (calls.count(0), calls.count(2), pieces.nit)
    n_calls = len(calls)
    monkeypatch.setenv("SX_CMA_PHASED", "0")
    whole = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="cmaes", options=dict(opts))
    assert len(calls) == n_calls
    assert (pieces.nit, pieces.nfev, pieces.status) == (whole.nit, whole.nfev, whole.status)
    assert pieces.fun == whole.fun and np.array_equal(pieces.x, whole.x)
    assert np.array_equal(pieces.xall, whole.xall) and np.array_equal(pieces.funall, whole.funall)
    if ftol > 0:
        assert pieces.status == 1 and pieces.nit < maxiter


@pytest.mark.parametrize("verbosity", [1.0, 0.4, 0.0])
def test_cmaes_device_resident_loop_history(sa, verbosity):
    """return_all without a callback stays on the device (history slabs written by a kernel, read back once): same
    shapes and values as the oracle's history (LAPACK + canonical signs)."""
    n, P = 12, 30
    opts = {"maxiter": 15, "popsize": P, "seed": 8, "sigma": 0.25, "return_all": True, "verbosity": verbosity}
    bounds = [[-2.0, 3.0]] * n
    ref = oracle.minimize("rosenbrock", bounds, method="cmaes", options=dict(opts, eigh="canonical"), rng="philox")
    got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="cmaes", options=dict(opts, backend="hip", rng="philox"))
    assert got.xall.shape == ref.xall.shape and got.funall.shape == ref.funall.shape
    assert np.allclose(got.funall, ref.funall, rtol=1e-6) and np.allclose(got.xall, ref.xall, rtol=1e-6, atol=1e-9)
    assert (got.nit, got.status) == (ref.nit, ref.status)


def test_cmaes_device_resident_loop_stop_rules(sa):
    """Stopping rules other than maxiter / ftol on the device: TolX-type stops on a flat objective region."""
    for obj, n, P, opts in (("sphere", 4, 8, {"maxiter": 3000, "ftol": -1.0, "xtol": 0.0, "sigma": 0.3}),
                            ("quartic", 5, 10, {"maxiter": 3000, "ftol": -1.0, "xtol": 0.0, "sigma": 0.3})):
        o = dict(opts, popsize=P, seed=5)
        ref = oracle.minimize(obj, [[-2.0, 2.0]] * n, method="cmaes", options=dict(o, eigh="canonical"), rng="philox")
        got = sa.optimize.minimize(getattr(sa.factory, obj), [[-2.0, 2.0]] * n, method="cmaes",
                                   options=dict(o, backend="hip", rng="philox"))
        assert ref.status < -1, ref.status  # one of the CMA-specific rules fired
        assert got.status == ref.status and abs(got.nit - ref.nit) <= max(3, ref.nit // 50)


def test_cmaes_device_eigensolver_converges(sa):
    """eigh="device" in the cases where the eigenbasis is NOT determined (mu + 1 < n: a repeated eigenvalue, any
    basis of its eigenspace is valid) -- the run must still behave like CMA-ES: converge on the sphere with
    status 1, and on the reference-suite problem."""
    res = sa.optimize.minimize(sa.factory.sphere, [[-5.12, 5.12]] * 12, method="cmaes",
                               options={"maxiter": 600, "popsize": 32, "seed": 3, "eigh": "device"})
    assert res.status == 1 and res.fun <= 1e-8 and np.abs(res.x).max() < 1e-3
    res = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * 2, method="cmaes",
                               options={"maxiter": 300, "popsize": 10, "seed": 0, "eigh": "device", "rng": "philox"})
    assert res.success and np.allclose(res.x, [1.0, 1.0], atol=1e-3)


@pytest.mark.parametrize("objective,n,P,maxiter", [("rosenbrock", 20, 20, 300), ("sphere", 8, 12, 119), ("rosenbrock", 6, 12, 60),
                                                   ("rastrigin", 40, 16, 80), ("sphere", 70, 10, 40),
                                                   ("rosenbrock", 512, 1024, 12)])  # BASELINE config 4's own size (round 4)
def test_device_loop_generation_by_generation_from_the_oracles_state(sa, ctx, objective, n, P, maxiter):
    """The device-resident CMA-ES generation (csrc/sx_cma_loop.hip) checked one generation at a time: every
    generation starts from the ORACLE's model of that generation (mean, paths, C, B, D, sigma, best-f history) and
    must arrive at the oracle's next model -- candidates, fitness, best row, mean, ps, pc, C, sigma, status -- to
    rounding.  The new eigenvectors are checked by what is determined about them (eigenvalues, reconstruction,
    orthonormality): with mu + 1 < n (four of the five shapes) C keeps a repeated eigenvalue, whose eigenspace has no
    canonical basis, so whole-run traces of two solvers part ways there -- this form of the test does not care.
    The last shape is BASELINE config 4 itself: there the eigenbasis IS determined, but its closest eigenvalue pairs are
    1e-9 ... 1e-6 of |C| apart in the first generations, so the eigenvectors of two solvers differ by rounding / gap and WHOLE
    runs stay within 1e-6 of each other only for a seed-dependent number of generations (profiles/r4_c4_parity_margin.txt:
    60+ for some seeds, 8 for others, with or without the refinement step) -- generation by generation every step is exact."""
    import torch

    from stochopy_amd import _lib
    from stochopy_amd.optimize._cmaes import _CmaDeviceRun

    seed, sigma0 = 4242, 0.3
    bounds = np.array([[-3.0, 4.0]] * n)
    steps = []
    oracle.minimize(objective, bounds, method="cmaes", rng="philox",
                    options=dict(maxiter=maxiter, popsize=P, sigma=sigma0, seed=seed, eigh="canonical", xtol=1e-12,
                                 ftol=1e-30, probe=lambda it, before, after: steps.append((it, before, after))))
    assert len(steps) >= min(maxiter, 30)
    run = _CmaDeviceRun(getattr(sa.factory, objective).sx_id, bounds[:, 0].copy(), bounds[:, 1].copy(), None, maxiter, P,
                        sigma0, 0.5, 1e-12, 1e-30, seed, run=False)
    buf = run.buffers
    mu = int(0.5 * P)

    def put(name, value):
        buf[name].copy_(torch.from_numpy(np.array(value, dtype=np.float64, order="C", copy=True)))

    def close(a, b, tol):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        return np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300)

    decomposed = 0
    with torch.cuda.stream(run.ctx.stream):
        for it, before, after in steps:
            for name in ("xmean", "ps", "pc", "C", "B", "D"):
                put(name, before[name])
            put("besthist", before["besthist"])
            st = _lib.SxCmaState(it=it - 1, nfev=(it - 1) * P, best_row=0, fbest=0.0, sigma=before["sigma"],
                                 sigma_next=before["sigma"], tmp_coef=0.0, psnorm=0.0, status=_lib.SX_STATUS_NONE, done=0,
                                 stop_it=0)
            put("state", np.frombuffer(bytes(st), dtype=np.float64))
            due = (1 + decomposed % 2) if after["due"] else 0  # cold start / started from the current B, alternately
            decomposed += bool(after["due"])
            run.step(it, due)
            got = run.read_state()
            best = int(after["order"][0])
            assert close(buf["arx"].cpu().numpy(), after["arx"], 1e-12), it
            assert np.allclose(buf["fit"].cpu().numpy(), after["arfit"], rtol=1e-11, atol=1e-300), it
            assert (got.it, got.nfev, got.best_row) == (it, it * P, best), it
            assert np.isclose(got.fbest, after["arfit"][best], rtol=1e-11, atol=0), it
            assert close(buf["xmean"].cpu().numpy(), after["xmean"], 1e-12), it
            assert close(buf["ps"].cpu().numpy(), after["ps"], 1e-10), it
            assert close(buf["pc"].cpu().numpy(), after["pc"], 1e-10), it
            Cgot = buf["C"].cpu().numpy()
            if not after["due"]:  # (the reference symmetrises only when it decomposes: compare the upper triangle)
                assert close(np.triu(Cgot), np.triu(after["C"]), 1e-11), it
            else:
                assert close(Cgot, after["C"], 1e-11), it
                Dg, Bg = buf["D"].cpu().numpy(), buf["B"].cpu().numpy()
                assert close(Dg, after["D"], 1e-10), it
                assert np.abs(Bg.T @ Bg - np.eye(n)).max() <= 1e-12, it
                assert close((Bg * Dg**2) @ Bg.T, after["C"], 1e-11), it
                k = np.abs(Bg).argmax(axis=0)  # canonical sign: the largest component of every eigenvector is positive
                assert (Bg[k, np.arange(n)] > 0).all(), it
            assert np.isclose(got.sigma, after["sigma"], rtol=1e-10, atol=0), it
            want = _lib.SX_STATUS_NONE if after["status"] is None else after["status"]
            assert (got.status, bool(got.done)) == (want, after["status"] is not None), it
    assert decomposed >= 10
