"""GPU parity: objectives and DE generations through the C ABI vs the oracle / golden vectors."""
import numpy as np
import pytest

import oracle
from oracle.objectives import OBJECTIVES
from conftest import case_bounds, load_golden, unhex

pytestmark = pytest.mark.gpu

# objectives made only of + - * reproduce numpy bit for bit; the others contain cos/exp/sqrt/pow
EXACT = {"rosenbrock", "sphere"}
RTOL_TRANSCENDENTAL = 1e-13


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


@pytest.mark.parametrize("n", [1, 2, 3, 7, 8, 9, 13, 16, 17, 64, 127, 128, 129, 130, 255, 256, 257, 1000, 1023, 1024, 1025, 2049, 3071, 4095,
                               4096])
@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_objectives_vs_oracle(sa, name, n):
    rs = np.random.RandomState(n * 7 + 1)
    X = rs.uniform(-5.12, 5.12, (37, n))
    got = getattr(sa.factory, name)(X)
    ref = OBJECTIVES[name](X)
    if name in EXACT:
        assert np.array_equal(got, ref)
    else:
        assert np.allclose(got, ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)


@pytest.mark.parametrize("n", [64, 128, 256])
@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_objectives_one_batch_rows_eight_lanes_per_row(sa, name, n):
    """Round 5: large populations of one-batch rows (n = 64 / 128 / 256, P >= 32768, a multiple of 32) are evaluated by
    eval_r8_kernel -- eight lanes per row, one per accumulator of numpy's sum, the tail terms added by a DPP scan, rows staged
    through 16-byte lane loads -- : the same bits as the oracle (and as the 16 / 32 / 64-lane kernel, which the same
    population minus 32 rows still takes: P % 32 != 0 ... here P = 32768 + 16 rows take it)."""
    import torch
    from stochopy_amd import _device, _lib

    rs = np.random.RandomState(n + 3)
    X = rs.uniform(-5.12, 5.12, (32768 + 16, n))
    ref = OBJECTIVES[name](X)
    ctx = _device.Context()
    Xd = torch.as_tensor(X, device=ctx.device)
    torch.cuda.synchronize()
    got8 = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd[:32768], n)  # eight lanes per row
    got = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd, n)            # P % 32 != 0: the one-visit kernel
    ctx.sync()
    got8, got = got8.cpu().numpy(), got.cpu().numpy()
    assert np.array_equal(got8, got[:32768])  # the two kernels agree bit for bit, whatever the objective
    if name in EXACT:
        assert np.array_equal(got, ref)
    else:
        assert np.allclose(got, ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)


# 16 / 17: one leaf of two blocks (+ a tail); 63 / 100 / 129: one leaf; 135 ... 200: two leaves; 249 / 250 / 255: three leaves
# (numpy cuts a right part of 129 ... 135 terms once more); 256 with P % 32 != 0: two full leaves
@pytest.mark.parametrize("n", [16, 17, 63, 100, 129, 135, 136, 200, 249, 250, 255, 256])
@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_objectives_one_batch_rows_of_any_length_eight_lanes_per_row(sa, name, n):
    """Round 5: large populations (P >= 32768) of one-batch rows OFF the compile-time grid are evaluated by eval_r8_rt_kernel --
    eight lanes per row straight from memory, numpy's plan (at most three leaves) at run time: the same bits as the oracle and
    as the 16 / 32 / 64-lanes-per-row kernel, which a smaller population of the same rows takes."""
    import torch
    from stochopy_amd import _device, _lib

    rs = np.random.RandomState(n + 5)
    X = rs.uniform(-5.12, 5.12, (32768 + 13, n))
    ref = OBJECTIVES[name](X)
    ctx = _device.Context()
    Xd = torch.as_tensor(X, device=ctx.device)
    torch.cuda.synchronize()
    got8 = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd, n)          # eight lanes per row, a last wavefront of 5 rows
    got = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd[:4099], n)    # fewer rows: the lanes-per-row kernel
    ctx.sync()
    got8, got = got8.cpu().numpy(), got.cpu().numpy()
    assert np.array_equal(got8[:4099], got)
    if name in EXACT:
        assert np.array_equal(got8, ref)
    else:
        assert np.allclose(got8, ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)


# 257: three leaves (129 -> 64 + 65 for Rosenbrock's 256 terms: two); 300 / 700 / 1023 / 1500: plans of 4 ... 16 leaves with tails;
# 2047: 17 leaves; 2000 with a row stride of 2001 doubles (rows 8-byte aligned only)
# 512 / 2048: the cosine objectives take this kernel there too (the cheap ones their compile-time plan); 2049 ... 4096: the smaller
# population goes through the one-workgroup-per-row kernel (the cheap objectives beyond 3584 elements: both populations do)
@pytest.mark.parametrize("n", [257, 300, 512, 700, 1023, 1500, 2000, 2047, 2048, 2049, 3000, 3584, 4095, 4096])
@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_objectives_long_rows_of_any_length_eight_lanes_per_row(sa, name, n):
    """Round 5: large populations (P >= 32768) of rows of 257 ... 4096 elements are evaluated by eval_r8_long_kernel -- eight
    lanes per row straight from memory, the leaves one after the other, the recursion's combines by one lane: the same bits
    as the oracle and as the wavefront-per-row / workgroup-per-row kernels, which a smaller population takes."""
    import torch
    from stochopy_amd import _device, _lib

    P = 32768 + 5
    rs = np.random.RandomState(n + 9)
    ld = n + 1 if n == 2000 else n
    Xh = rs.uniform(-5.12, 5.12, (P, ld))
    ref = OBJECTIVES[name](Xh[:2051, :n])
    ctx = _device.Context()
    Xd = torch.as_tensor(Xh, device=ctx.device)
    torch.cuda.synchronize()
    got8 = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd[:, :n], n)        # eight lanes per row
    got = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd[:2051, :n], n)     # fewer rows: one wavefront per row
    ctx.sync()
    got8, got = got8.cpu().numpy(), got.cpu().numpy()
    assert np.array_equal(got8[:2051], got)
    if name in EXACT:
        assert np.array_equal(got, ref)
    else:
        assert np.allclose(got, ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)
    # the last rows too (a last workgroup of 5 rows), against the oracle
    tail_ref = OBJECTIVES[name](Xh[P - 37:, :n])
    if name in EXACT:
        assert np.array_equal(got8[P - 37:], tail_ref)
    else:
        assert np.allclose(got8[P - 37:], tail_ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)


@pytest.mark.parametrize("n", [512, 1024, 2048])
@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_objectives_long_rows_compile_time_plan(sa, name, n):
    """Whole batches of rows of 512 / 1024 / 2048 elements take the summation plan as compile-time constants
    (row_reduce_long: sx_eval directly, the generation kernels by a branch on the length): same bits as the oracle, with
    and without CMA-ES's affine map (cmaes/_cmaes.py:171)."""
    import torch
    from stochopy_amd import _device, _lib

    P = 16384 + 8 * 37
    rs = np.random.RandomState(n + 11)
    X = rs.uniform(-5.12, 5.12, (P, n))
    got = getattr(sa.factory, name)(X)
    ref = OBJECTIVES[name](X)
    if name in EXACT:
        assert np.array_equal(got, ref)
    else:
        assert np.allclose(got, ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)
    ctx = _device.Context()
    xm, xstd = rs.uniform(-1, 1, n), rs.uniform(0.5, 2.0, n)
    Xd = torch.as_tensor(X, device=ctx.device)
    f = _device.evaluate(ctx, _lib.FUN_IDS[name], Xd, n, xm=torch.as_tensor(xm, device=ctx.device),
                         xstd=torch.as_tensor(xstd, device=ctx.device))
    ctx.sync()
    ref = OBJECTIVES[name](X * xstd + xm)
    if name in EXACT:
        assert np.array_equal(f.cpu().numpy(), ref)
    else:
        assert np.allclose(f.cpu().numpy(), ref, rtol=RTOL_TRANSCENDENTAL, atol=1e-13)


def test_objective_known_answers(sa):
    """reference tests/test_factory.py:7-23"""
    refs = load_golden("factory_kat.json")["test_factory_refs"]
    for name, ref in refs.items():
        assert np.allclose(ref, getattr(sa.factory, name)(np.ones(10)))


def _run_hip(sa, case, rng="numpy-legacy", **extra):
    trace = []
    opts = dict(case["options"])
    opts.update({"backend": "hip", "rng": rng})
    opts.update(extra)
    fun = getattr(sa.factory, case["objective"])
    res = sa.optimize.minimize(fun, case_bounds(case), x0=case["x0"], method=case["method"], options=opts,
                               callback=lambda X, r: trace.append((float(r.fun), X.copy())))
    return res, trace


DE_CASES = [c for c in load_golden("configs.json")["cases"] if c["method"] == "de"]


@pytest.mark.parametrize("case", DE_CASES, ids=lambda c: c["tag"])
def test_de_matches_reference_golden(sa, case):
    """numpy-legacy stream: same seed => the reference's per-generation best-f, x, nit, status."""
    res, trace = _run_hip(sa, case)
    ref = case["result"]
    got = np.array([t[0] for t in trace])
    want = unhex(case["fun_trace"])
    if case["objective"] in EXACT:
        assert np.array_equal(got, want)
        assert float(res.fun).hex() == ref["fun"]
        for g, rows in case["pop_rows"].items():
            for r, row in enumerate(rows):
                assert np.array_equal(unhex(row), trace[int(g)][1][r, : len(row)])
    else:
        assert np.allclose(got, want, rtol=1e-6, atol=0)  # north-star tolerance: best-f within 1e-6 rel
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])


def test_de_c2_long_run_follows_the_reference(sa):
    """BASELINE config 2 (Rastrigin -- device cos) over 40 generations at full size: best-f of every generation, the final
    x and the whole population (projection every 5 generations) against the reference's (a flipped
    `<` deep in the population must not pass)."""
    from conftest import check_long_case

    case = {c["tag"]: c for c in load_golden("configs_long.json")["cases"]}["C2L_de_rastrigin_n128_p4096"]
    check_long_case(sa, case)


@pytest.mark.parametrize("tag", ["de_rand1bin", "de_rand2bin", "de_best1bin", "de_best2bin", "de_rand1bin_random"])
def test_de_reference_suite_xrefs(sa, tag):
    """The reference's own xrefs (tests/test_optimize.py:51-86, deferred rows) incl. return_all."""
    import os
    from conftest import GOLDEN

    case = {c["tag"]: c for c in load_golden("suite_rosen2d.json")["cases"]}[tag]
    res, _ = _run_hip(sa, case)
    assert np.allclose(case["xref_from_reference_tests"], res.x)
    arrays = np.load(os.path.join(GOLDEN, "suite_rosen2d_xall.npz"))
    assert np.array_equal(arrays[tag + "__xall"], res.xall)
    assert np.array_equal(arrays[tag + "__funall"], res.funall)


@pytest.mark.parametrize("strategy", ["rand1bin", "rand2bin", "best1bin", "best2bin"])
@pytest.mark.parametrize("constraints", [None, "Random"])
@pytest.mark.parametrize("shape", [(5, 12), (37, 100), (128, 256), (300, 64)])
def test_de_philox_matches_oracle(sa, strategy, constraints, shape):
    """Philox mode: device draws == oracle/streams.py PhiloxStream, so traces are bit-identical."""
    n, P = shape
    opts = {"maxiter": 8, "popsize": P, "seed": 1234567 + n, "strategy": strategy, "constraints": constraints,
            "mutation": 0.7, "recombination": 0.6, "updating": "deferred"}
    bounds = [[-2.0, 2.0]] * n
    t_ref, t_got = [], []
    r_ref = oracle.minimize("rosenbrock", bounds, method="de", options=dict(opts), rng="philox",
                            callback=lambda X, r: t_ref.append((r.fun, X.copy())))
    o = dict(opts, backend="hip", rng="philox")
    r_got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=o,
                                 callback=lambda X, r: t_got.append((r.fun, X.copy())))
    assert len(t_ref) == len(t_got)
    for (fa, Xa), (fb, Xb) in zip(t_ref, t_got):
        assert fa == fb
        assert np.array_equal(Xa, Xb)
    assert np.array_equal(r_ref.x, r_got.x) and r_ref.nit == r_got.nit and r_ref.status == r_got.status


def test_de_graph_equals_stepwise(sa):
    """hipGraph replay (no callback) must give the same result as per-generation launches."""
    n, P = 128, 4096
    bounds = [[-5.12, 5.12]] * n
    o = {"maxiter": 130, "popsize": P, "seed": 5, "updating": "deferred", "backend": "hip", "rng": "philox",
         "ftol": -1.0, "xtol": 0.0}
    a = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o))
    b = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o), callback=lambda X, r: None)
    assert a.nit == b.nit == 130 and a.fun == b.fun and np.array_equal(a.x, b.x)
    ref = oracle.minimize("rosenbrock", bounds, method="de", options={k: v for k, v in o.items() if k not in ("backend", "rng")} | {"maxiter": 12}, rng="philox")
    c = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o, maxiter=12))
    assert ref.fun == c.fun and np.array_equal(ref.x, c.x)


@pytest.mark.parametrize("objective", ["rosenbrock", "sphere"])
@pytest.mark.parametrize("shape", [(64, 64), (128, 256), (256, 64), (128, 4096)])
def test_de_one_batch_rows_through_replayed_graphs_match_oracle(sa, objective, shape):
    """Rows of exactly 64 / 128 / 256 elements with a population that fills whole workgroups take the chained kernel in
    which the row length -- and numpy's summation plan with it -- is a compile-time constant (csrc/sx_de.hip NFIX,
    sx_device.hpp row_reduce_fixed / pairwise_static).  No callback: the generations run as replayed graphs of that
    kernel.  Bit-identical to the oracle for every strategy."""
    n, P = shape
    for strategy, constraints in (("best1bin", None), ("rand1bin", "Random"), ("best2bin", None), ("rand2bin", None)):
        opts = {"maxiter": 120 if P <= 256 else 12, "popsize": P, "seed": 77 + n, "strategy": strategy, "constraints": constraints,
                "mutation": 0.6, "recombination": 0.8, "updating": "deferred"}
        bounds = [[-2.0, 2.0]] * n
        ref = oracle.minimize(objective, bounds, method="de", options=dict(opts), rng="philox")
        got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="de",
                                   options=dict(opts, backend="hip", rng="philox"))
        assert got.fun == ref.fun and np.array_equal(got.x, ref.x) and (got.nit, got.status) == (ref.nit, ref.status), strategy


@pytest.mark.parametrize("P,n", [(2, 3), (7, 5), (100, 70), (513, 2), (4096, 128), (16384, 256), (1000, 1030)])
def test_philox_latin_hypercube_kernel_vs_oracle(sa, P, n):
    """sx_philox_lhs (the initial population of rng="philox" runs, drawn on the device) == oracle
    PhiloxStream.lhs_population bit for bit, whole and in shards (row0 / rows as a rank of a sharded run uses them)."""
    from stochopy_amd import _device, _rng

    ctx = _device.Context()
    seed = 2**40 + 12345 + P
    lo, up = np.full(n, -5.12), np.linspace(1.0, 5.12, n)
    want = oracle.PhiloxStream(seed).lhs_population(P, n, lo, up)
    d_lo, d_up = ctx.upload(lo), ctx.upload(up)
    got = _rng.philox_latin_hypercube(ctx, ctx.empty((P, n)), 0, P, d_lo, d_up, seed)
    ctx.sync()
    assert np.array_equal(got.cpu().numpy(), want)
    if P >= 7:
        r0, rows = P // 3, P // 2
        part = _rng.philox_latin_hypercube(ctx, ctx.empty((rows, n)), r0, P, d_lo, d_up, seed)
        ctx.sync()
        assert np.array_equal(part.cpu().numpy(), want[r0 : r0 + rows])
