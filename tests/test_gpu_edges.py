"""GPU edge cases the reference's own tests touch (tests/helpers.py, tests/test_optimize.py) or imply:
x0 handling, verbosity=0 history, maxiter <= 1, callback count, ragged / tiny / large dimensions."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


B2 = [[-5.12, 5.12]] * 2


@pytest.mark.parametrize("method", ["de", "pso", "cpso", "cmaes"])
def test_callback_count_equals_maxiter(sa, method):
    """reference tests/test_optimize.py:135-152: the callback fires exactly maxiter times."""
    for maxiter in (2, 5, 9):
        count = []
        sa.optimize.minimize(sa.factory.rosenbrock, B2, method=method, options={"maxiter": maxiter, "seed": 1},
                             callback=lambda X, s: count.append(X.shape))
        assert len(count) == maxiter
        assert all(shape == (10, 2) for shape in count)


@pytest.mark.parametrize("method", ["de", "pso"])
def test_x0_population_and_inplace_semantics(sa, method):
    rs = np.random.RandomState(3)
    x0 = rs.uniform(-5, 5, (16, 3))
    keep = x0.copy()
    opts = {"maxiter": 12, "popsize": 16, "seed": 4, "updating": "deferred"}
    ref = oracle.minimize("sphere", [[-5.12, 5.12]] * 3, x0=keep.copy(), method=method, options=dict(opts))
    got = sa.optimize.minimize(sa.factory.sphere, [[-5.12, 5.12]] * 3, x0=x0, method=method, options=dict(opts, backend="hip"))
    assert got.fun == ref.fun and np.array_equal(got.x, ref.x) and got.nit == ref.nit
    if method == "de":  # the reference works in place on x0 (de/_de.py:208): the final population comes back in it
        assert not np.array_equal(x0, keep)
    else:               # sync PSO rebinds X (cpso/_constraints.py:8): x0 untouched
        assert np.array_equal(x0, keep)
    with pytest.raises(ValueError):
        sa.optimize.minimize(sa.factory.sphere, [[-5.12, 5.12]] * 3, x0=keep[:5], method=method, options=opts)


@pytest.mark.parametrize("method", ["de", "pso", "cmaes"])
@pytest.mark.parametrize("verbosity", [0.0, 0.3, 1.0])
def test_return_all_shapes_and_values(sa, method, verbosity):
    opts = {"maxiter": 7, "popsize": 10, "seed": 2, "return_all": True, "verbosity": verbosity, "updating": "deferred"}
    if method == "cmaes":
        opts.pop("updating")
    ref = oracle.minimize("rosenbrock", B2, method=method, options=dict(opts))
    got = sa.optimize.minimize(sa.factory.rosenbrock, B2, method=method, options=dict(opts, backend="hip"))
    rows = max(int(np.ceil(verbosity * 10)), 1)
    assert got.xall.shape == (got.nit, rows, 2) and got.funall.shape == (got.nit, rows)
    if method == "cmaes":  # MFMA vs BLAS summation order: north-star tolerance instead of bit equality
        assert np.allclose(got.xall, ref.xall, rtol=1e-6, atol=1e-9) and np.allclose(got.funall, ref.funall, rtol=1e-6)
    else:
        assert np.array_equal(got.xall, ref.xall) and np.array_equal(got.funall, ref.funall)


@pytest.mark.parametrize("method,objective,gens", [("de", "rosenbrock", 75), ("pso", "rosenbrock", 75),
                                                   ("cpso", "sphere", 45), ("de", "sphere", 400)])
def test_return_all_philox_streams_history_without_host_round_trips(sa, method, objective, gens):
    """Philox draws + return_all: the history slabs are copied device-side on the engine stream and the host
    looks at the state every 32 generations only -- same xall / funall / nit as the oracle, including runs that
    converge between two looks (DE on Sphere stops on ftol) and CPSO restarts."""
    n, P = 12, 64
    b = [[-5.12, 5.12]] * n
    opts = {"maxiter": gens, "popsize": P, "seed": 21, "return_all": True, "verbosity": 0.5, "updating": "deferred",
            "ftol": 1e-6 if objective == "sphere" and method == "de" else -1.0}
    ref = oracle.minimize(objective, b, method=method, options=dict(opts), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, objective), b, method=method,
                               options=dict(opts, backend="hip", rng="philox"))
    assert (got.nit, got.status, got.fun) == (ref.nit, ref.status, ref.fun)
    assert got.xall.shape == (ref.nit, P // 2, n)
    assert np.array_equal(got.xall, ref.xall) and np.array_equal(got.funall, ref.funall)


@pytest.mark.parametrize("method", ["de", "pso"])
def test_maxiter_one_still_runs_a_generation(sa, method):
    """`it` starts at 1 and is incremented before the test `it >= maxiter` (de/_de.py:245-247)."""
    r = sa.optimize.minimize(sa.factory.sphere, B2, method=method, options={"maxiter": 1, "popsize": 8, "seed": 0})
    assert r.nit == 2 and r.nfev == 16 and r.status == -1 and r.success is False
    for rng in ("numpy-legacy", "philox"):
        r = sa.optimize.minimize(sa.factory.sphere, B2, method=method,
                                 options={"maxiter": 2, "popsize": 8, "seed": 0, "rng": rng})
        assert r.nit == 2 and r.status == -1


def test_early_termination_statuses_match_oracle_in_philox_mode(sa):
    """ftol / xtol ladder (status 1 and 0) in the one-kernel-per-generation path, where the host settles 0 vs 1."""
    b = [[-5.12, 5.12]] * 4
    for xtol, want in ((1e-3, None), (10.0, 0)):
        opts = {"maxiter": 400, "popsize": 64, "seed": 9, "ftol": 1e-6, "xtol": xtol, "updating": "deferred"}
        ref = oracle.minimize("sphere", b, method="de", options=dict(opts), rng="philox")
        got = sa.optimize.minimize(sa.factory.sphere, b, method="de", options=dict(opts, backend="hip", rng="philox"))
        assert (got.nit, got.status, got.fun, got.message) == (ref.nit, ref.status, ref.fun, ref.message)
        assert np.array_equal(got.x, ref.x)
        if want is not None:
            assert got.status == want


@pytest.mark.parametrize("n,P", [(1, 8), (2, 6), (65, 33), (129, 70), (257, 40), (1000, 24), (2560, 12), (4095, 9),
                                 (4096, 10)])
def test_ragged_shapes_philox_de_and_pso(sa, n, P):
    b = [[-3.0, 3.0]] * n
    for method in ("de", "pso"):
        opts = {"maxiter": 5, "popsize": P, "seed": 5 + n, "updating": "deferred", "constraints": None}
        ref = oracle.minimize("rosenbrock", b, method=method, options=dict(opts), rng="philox")
        got = sa.optimize.minimize(sa.factory.rosenbrock, b, method=method, options=dict(opts, backend="hip", rng="philox"))
        assert got.fun == ref.fun and np.array_equal(got.x, ref.x), (method, n, P)


@pytest.mark.parametrize("objective", ["rosenbrock", "rastrigin", "griewank"])
@pytest.mark.parametrize("n,P", [(513, 40), (1024, 64), (2048, 24)])
def test_long_rows_fused_reduction(sa, objective, n, P):
    """n > 256: the objective terms are formed inside the numpy-order reduction (no term arrays in LDS);
    DE (two-kernel path, Shrink-free PSO, Shrink PSO) must still follow the oracle: bit for bit for +,-,*
    objectives, best-f within 1e-12 rel where cos/sqrt are involved."""
    b = [[-3.0, 3.0]] * n
    for method, cons in (("de", None), ("de", "Random"), ("pso", None), ("pso", "Shrink")):
        opts = {"maxiter": 6, "popsize": P, "seed": 11 + n, "updating": "deferred", "constraints": cons}
        ref = oracle.minimize(objective, b, method=method, options=dict(opts), rng="philox")
        got = sa.optimize.minimize(getattr(sa.factory, objective), b, method=method,
                                   options=dict(opts, backend="hip", rng="philox"))
        if objective == "rosenbrock":
            assert got.fun == ref.fun and np.array_equal(got.x, ref.x), (method, cons)
        else:
            assert np.isclose(got.fun, ref.fun, rtol=1e-12, atol=0), (method, cons, got.fun, ref.fun)


def test_c5_shard_shape_three_generations(sa):
    """BASELINE config 5's per-GPU shard (DE, n=1024, P=16384) at full size: three generations on the device
    == the oracle, population rows included (bit-exact; Rosenbrock is +,-,* only)."""
    n, P = 1024, 16384
    b = [[-5.12, 5.12]] * n
    opts = {"maxiter": 4, "popsize": P, "seed": 3, "updating": "deferred", "ftol": -1.0, "xtol": 0.0}
    t_ref, t_got = [], []
    pick = np.array([0, 1, 4095, 8192, 16383])
    ref = oracle.minimize("rosenbrock", b, method="de", options=dict(opts), rng="philox",
                          callback=lambda X, r: t_ref.append((r.fun, X[pick].copy())))
    got = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(opts, backend="hip", rng="philox"),
                               callback=lambda X, r: t_got.append((r.fun, X[pick].copy())))
    assert len(t_ref) == len(t_got) == 4
    for (fa, Xa), (fb, Xb) in zip(t_ref, t_got):
        assert fa == fb and np.array_equal(Xa, Xb)
    assert got.fun == ref.fun and np.array_equal(got.x, ref.x) and got.nfev == ref.nfev == 4 * P
    # and the asynchronous (no callback) path at the same size
    fast = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(opts, backend="hip", rng="philox"))
    assert fast.fun == ref.fun and np.array_equal(fast.x, ref.x)


def test_c5_full_population_properties(sa):
    """BASELINE config 5 at FULL size on one GPU (DE, n=1024, P=131072: two 1 GiB population buffers), where the
    oracle is out of reach: size-independent properties instead -- the run is deterministic (two runs, same bits),
    the best value never increases and is the objective of the returned row, counters add up, and the history
    (verbosity 0: the best candidate of each generation) never beats the running best."""
    n, P, gens = 1024, 131072, 6
    b = [[-5.12, 5.12]] * n
    opts = {"maxiter": gens, "popsize": P, "seed": 17, "ftol": -1.0, "xtol": 0.0, "backend": "hip", "rng": "philox",
            "updating": "deferred"}
    a1 = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(opts))
    trace = []
    a2 = sa.optimize.minimize(sa.factory.rosenbrock, b, method="de", options=dict(opts, return_all=True, verbosity=0.0),
                              callback=lambda X, r: trace.append((float(r.fun), X.shape, int(r.nfev))))
    assert a1.fun == a2.fun and np.array_equal(a1.x, a2.x)                      # graph replay == stepwise, same bits
    assert (a1.nit, a1.nfev, a1.status) == (gens, gens * P, -1)
    assert a1.fun == sa.factory.rosenbrock(a1.x)                                # the value of the returned row
    f = np.array([t[0] for t in trace])
    assert len(trace) == gens and np.all(np.diff(f) <= 0.0) and f[-1] == a1.fun
    assert all(t[1] == (P, n) and t[2] == (k + 1) * P for k, t in enumerate(trace))
    assert a2.xall.shape == (gens, 1, n) and np.all(a2.funall[1:, 0] >= f[1:])  # a candidate cannot beat the best kept
    assert np.all(np.abs(a2.xall) < 5.12 * 3)                                   # best1bin without repair stays near the box


def test_result_surface(sa):
    r = sa.optimize.minimize(sa.factory.rosenbrock, B2, method="cmaes", options={"maxiter": 100, "popsize": 10, "seed": 0})
    # README example of the reference (README.rst:93-105): nit 49, nfev 490, status 1
    assert (r.nit, r.nfev, r.status, r.success) == (49, 490, 1, True)
    assert np.allclose(r.x, [0.99997096, 0.99993643]) and np.isclose(r.fun, 3.862267664744548e-09, rtol=1e-6)
    assert sorted(r.keys()) == ["fun", "message", "nfev", "nit", "status", "success", "x"]
    assert r.message == "best solution value is lower than ftol"


def test_concurrent_calls_from_two_host_threads(sa):
    """SURVEY.md section 8 row b6 (threading / streams): every host thread gets its own engine stream (kept for the life of
    the process, so that torch's caching allocator can hand a run the previous run's buffers back) -- two threads calling
    minimize() at the same time, repeatedly and with different methods, get the results of the same calls made one
    after the other."""
    import threading

    jobs = {
        "de": (sa.factory.rosenbrock, [[-5.12, 5.12]] * 40, "de",
               {"maxiter": 60, "popsize": 256, "seed": 3, "rng": "philox", "updating": "deferred", "backend": "hip"}),
        "pso": (sa.factory.ackley, [[-5.12, 5.12]] * 64, "pso",
                {"maxiter": 50, "popsize": 512, "seed": 4, "rng": "philox", "updating": "deferred", "backend": "hip"}),
        "cmaes": (sa.factory.rosenbrock, [[-3.0, 3.0]] * 12, "cmaes",
                  {"maxiter": 40, "popsize": 24, "seed": 5, "rng": "philox", "backend": "hip"}),
    }

    def run(tag):
        fun, bounds, method, opts = jobs[tag]
        return sa.optimize.minimize(fun, bounds, method=method, options=dict(opts))

    serial = {tag: run(tag) for tag in jobs}
    for pair in (("de", "pso"), ("cmaes", "de"), ("pso", "cmaes")):
        out, errs = {}, []

        def work(tag):
            try:
                for _ in range(3):
                    out[tag] = run(tag)
            except Exception as exc:  # noqa: BLE001 -- reported below
                errs.append((tag, exc))

        threads = [threading.Thread(target=work, args=(tag,)) for tag in pair]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        assert not errs, errs
        for tag in pair:
            assert (out[tag].nit, out[tag].status) == (serial[tag].nit, serial[tag].status)
            assert out[tag].fun == serial[tag].fun and np.array_equal(out[tag].x, serial[tag].x)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["de", "pso", "cpso", "na"])
def test_workers_with_the_references_own_stream_do_not_change_the_result(sa, method):
    """The reference's invariant (tests/helpers.py:28-36, stochopy/optimize/_common.py:58-72): the parallel backend must not
    change the result for a seed.  With rng="numpy-legacy" (one sequential host stream) and for NA a run cannot be sharded: it
    runs replicated -- workers=3 returns the workers=1 result bit for bit and says so once -- where rounds 1-5 raised."""
    import warnings

    from stochopy_amd.optimize import _common

    _common._warned_replicated.discard(method)
    opts = {"maxiter": 15, "popsize": 24, "seed": 9, "rng": "numpy-legacy", "backend": "hip"}
    if method != "na":
        opts["updating"] = "deferred"  # (what the reference itself runs with a parallel backend: de/_de.py:142-145)
    bounds = [[-5.12, 5.12]] * 8
    one = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method=method, options=dict(opts))
    with pytest.warns(UserWarning, match="replicated"):
        three = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method=method, options=dict(opts, workers=3))
    with warnings.catch_warnings():
        warnings.simplefilter("error", UserWarning)  # (once per method)
        again = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method=method, options=dict(opts, workers=3))
    for r in (three, again):
        assert r.fun == one.fun and np.array_equal(r.x, one.x) and (r.nit, r.nfev, r.status) == (one.nit, one.nfev, one.status)
