"""GPU parity of the BENCHMARKED configurations in the benchmarked mode: BASELINE configs 2 / 3a / 3b at full size with in-kernel
Philox draws and no callback -- replayed graphs of the fused kernels, exactly what bench.py times -- against the oracle's
PhiloxStream run, through the WHOLE population of every generation (return_all, verbosity 1: the history is kept in HBM)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


@pytest.mark.parametrize("method,objective,n,P,gens,extra", [
    ("de", "rastrigin", 128, 4096, 25, {"strategy": "best1bin"}),     # config 2 (device cos)
    ("de", "rosenbrock", 128, 4096, 25, {"strategy": "best1bin"}),    # the metric shape: bit for bit
    ("pso", "ackley", 256, 16384, 6, {}),                             # config 3a (device cos / exp / sqrt)
    ("cpso", "ackley", 256, 16384, 8, {}),                            # config 3b: the competitive restart fires
])
def test_benchmarked_configs_in_philox_mode_follow_the_oracle(sa, method, objective, n, P, gens, extra):
    opts = dict({"maxiter": gens, "popsize": P, "seed": 0, "updating": "deferred", "ftol": -1.0, "xtol": 0.0,
                 "return_all": True, "verbosity": 1.0}, **extra)
    bounds = [[-5.12, 5.12]] * n
    ref = oracle.minimize(objective, bounds, method=method, options=dict(opts), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method=method, options=dict(opts, backend="hip", rng="philox"))
    assert (got.nit, got.nfev, got.status) == (ref.nit, ref.nfev, ref.status)
    assert got.xall.shape == ref.xall.shape == (gens, P, n)
    if objective in ("rosenbrock", "sphere"):  # +, -, * only: the same bits
        assert got.fun == ref.fun and np.array_equal(got.x, ref.x)
        assert np.array_equal(got.xall, ref.xall) and np.array_equal(got.funall, ref.funall)
    else:
        # the positions are made of +, -, * of identical draws: any difference is a `<` that went the other way somewhere
        for g in range(gens):
            bad = np.flatnonzero(np.abs(got.xall[g] - ref.xall[g]).max(axis=1) > 1e-6 * 10.24)
            assert bad.size == 0, f"generation {g + 1}: rows {bad[:8]} differ"
        assert np.allclose(got.funall, ref.funall, rtol=1e-9, atol=0)  # device cos / exp vs libm: a few ulp per term
        assert np.isclose(got.fun, ref.fun, rtol=1e-6, atol=0) and np.allclose(got.x, ref.x, rtol=0, atol=1e-6 * 10.24)
    if method == "cpso":
        assert len(ref["_restarts"]) >= 3  # the restart fired in the oracle's run (and the populations above agree)
