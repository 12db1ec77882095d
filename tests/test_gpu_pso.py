"""GPU parity: PSO / CPSO generations through the C ABI vs golden vectors and the oracle."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, case_bounds, load_golden, unhex

pytestmark = pytest.mark.gpu
EXACT = {"rosenbrock", "sphere"}


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


def _run_hip(sa, case, rng="numpy-legacy", **extra):
    trace = []
    opts = dict(case["options"])
    opts.update({"backend": "hip", "rng": rng})
    opts.update(extra)
    fun = getattr(sa.factory, case["objective"])
    res = sa.optimize.minimize(fun, case_bounds(case), x0=case["x0"], method=case["method"], options=opts,
                               callback=lambda X, r: trace.append((float(r.fun), X.copy())))
    return res, trace


CASES = [c for c in load_golden("configs.json")["cases"] if c["method"] in ("pso", "cpso")]


@pytest.mark.parametrize("case", CASES, ids=lambda c: c["tag"])
def test_pso_matches_reference_golden(sa, case):
    """numpy-legacy stream: same seed => the reference's per-generation best-f, nit, status."""
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
        assert np.allclose(got, want, rtol=1e-6, atol=0)  # north-star tolerance
    assert (res.nit, res.nfev, res.status, res.success, res.message) == (
        ref["nit"], ref["nfev"], ref["status"], ref["success"], ref["message"])


@pytest.mark.parametrize("tag", ["C3aL_pso_ackley_n256_p16384", "C3bL_cpso_ackley_n256_p16384"])
def test_pso_c3_long_run_follows_the_reference(sa, tag, monkeypatch):
    """BASELINE configs 3a / 3b (Ackley -- device cos / exp / sqrt) over 30 generations at full size: best-f of every
    generation, the final x, the whole swarm (projection every 5 generations), and for CPSO the competitive restart's
    bookkeeping -- in which generations it fires, nw, and exactly WHICH rows it re-seeds (cpso/_cpso.py:405-426; the
    reference restarts ~16 000 of the 16 384 particles per generation here) -- against the reference's."""
    from conftest import check_long_case
    from stochopy_amd.optimize import _cpso

    case = {c["tag"]: c for c in load_golden("configs_long.json")["cases"]}[tag]
    fired = {}
    orig = _cpso._PsoRun._restart_host_order

    def spy(self, it):
        orig(self, it)
        rows = np.flatnonzero(self.pbestfit.cpu().numpy() == 1.0e30)
        if rows.size:
            fired[int(it)] = rows.astype(np.int32)

    monkeypatch.setattr(_cpso._PsoRun, "_restart_host_order", spy)
    check_long_case(sa, case, restart_rows=fired)


def test_cpso_population_history_bit_exact(sa):
    arrays = np.load(os.path.join(GOLDEN, "configs_pops.npz"))
    tag = "cpso_Shrink_rosenbrock_n16_p256"
    case = {c["tag"]: c for c in CASES}[tag]
    res, trace = _run_hip(sa, case)
    assert np.array_equal(arrays[tag + "__pops"], np.array([t[1] for t in trace]))


@pytest.mark.parametrize("tag", ["pso_none", "pso_shrink", "cpso_none", "cpso_shrink"])
def test_pso_reference_suite_xrefs(sa, tag):
    """The reference's own xrefs (tests/test_optimize.py:23-48, 95-118, deferred rows) incl. return_all."""
    case = {c["tag"]: c for c in load_golden("suite_rosen2d.json")["cases"]}[tag]
    res, _ = _run_hip(sa, case)
    assert np.allclose(case["xref_from_reference_tests"], res.x)
    arrays = np.load(os.path.join(GOLDEN, "suite_rosen2d_xall.npz"))
    assert np.array_equal(arrays[tag + "__xall"], res.xall)
    assert np.array_equal(arrays[tag + "__funall"], res.funall)
    if case["options"].get("constraints"):
        assert np.all(res.xall + 1.0e-15 >= -5.12) and np.all(res.xall - 1.0e-15 <= 5.12)


@pytest.mark.parametrize("method", ["pso", "cpso"])
@pytest.mark.parametrize("constraints", [None, "Shrink"])
@pytest.mark.parametrize("shape", [(5, 12), (37, 100), (130, 256), (300, 64), (64, 40), (128, 33), (256, 70)])
def test_pso_philox_matches_oracle(sa, method, constraints, shape):
    """Philox mode: device draws == oracle PhiloxStream; +,-,* objective => bit-identical traces."""
    n, P = shape
    opts = {"maxiter": 10, "popsize": P, "seed": 99 + n, "constraints": constraints, "updating": "deferred"}
    bounds = [[-2.0, 2.0]] * n
    t_ref, t_got = [], []
    r_ref = oracle.minimize("rosenbrock", bounds, method=method, options=dict(opts), rng="philox",
                            callback=lambda X, r: t_ref.append((r.fun, X.copy())))
    r_got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method=method,
                                 options=dict(opts, backend="hip", rng="philox"),
                                 callback=lambda X, r: t_got.append((r.fun, X.copy())))
    assert len(t_ref) == len(t_got)
    for (fa, Xa), (fb, Xb) in zip(t_ref, t_got):
        assert fa == fb
        assert np.array_equal(Xa, Xb)
    assert np.array_equal(r_ref.x, r_got.x) and r_ref.nit == r_got.nit and r_ref.status == r_got.status


@pytest.mark.parametrize("objective", ["sphere", "rosenbrock"])
@pytest.mark.parametrize("shape", [(64, 40), (128, 33), (256, 70), (256, 2048)])
def test_pso_whole_batch_rows_match_oracle(sa, objective, shape):
    """Rows of exactly 64 / 128 / 256 elements take the kernel in which the row length -- and with it numpy's summation
    plan -- is a compile-time constant (csrc/sx_pso.hip FULL, sx_device.hpp row_reduce_fixed): bit-identical to the
    oracle for the +,-,* objectives (one term per element: row_reduce_fixed; n - 1 terms: pairwise_static), PSO and CPSO
    (Shrink for the larger swarm)."""
    n, P = shape
    for method in ("pso", "cpso"):
        opts = {"maxiter": 25, "popsize": P, "seed": 5 + n, "updating": "deferred",
                "constraints": "Shrink" if P > 1000 else None}
        bounds = [[-2.0, 2.0]] * n
        ref = oracle.minimize(objective, bounds, method=method, options=dict(opts), rng="philox")
        got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method=method,
                                   options=dict(opts, backend="hip", rng="philox"))
        assert got.fun == ref.fun and np.array_equal(got.x, ref.x) and (got.nit, got.status) == (ref.nit, ref.status)


def test_cpso_restart_selection_above_32768_particles(sa):
    """The one-workgroup radix selection keeps its keys in registers up to 32768 particles and re-reads them per
    pass above that: a swarm of 40000 whose restarts fire every generation must still follow the oracle."""
    n, P = 8, 40000
    opts = {"maxiter": 6, "popsize": P, "seed": 13, "updating": "deferred"}
    bounds = [[-5.12, 5.12]] * n
    ref = oracle.minimize("sphere", bounds, method="cpso", options=dict(opts), rng="philox")
    assert len(ref["_restarts"]) >= 3
    got = sa.optimize.minimize(sa.factory.sphere, bounds, method="cpso", options=dict(opts, backend="hip", rng="philox"))
    assert got.fun == ref.fun and np.array_equal(got.x, ref.x) and got.nit == ref.nit


def test_cpso_philox_restarts_fire_and_match_oracle(sa):
    """Ackley n16 P256: restarts fire; device selection must reset the same rows."""
    n, P = 16, 256
    opts = {"maxiter": 30, "popsize": P, "seed": 5, "updating": "deferred"}
    bounds = [[-5.12, 5.12]] * n
    t_ref, t_got = [], []
    r_ref = oracle.minimize("sphere", bounds, method="cpso", options=dict(opts), rng="philox",
                            callback=lambda X, r: t_ref.append((r.fun, X.copy())))
    assert len(r_ref["_restarts"]) > 0
    r_got = sa.optimize.minimize(sa.factory.sphere, bounds, method="cpso", options=dict(opts, backend="hip", rng="philox"),
                                 callback=lambda X, r: t_got.append((r.fun, X.copy())))
    for (fa, Xa), (fb, Xb) in zip(t_ref, t_got):
        assert fa == fb and np.array_equal(Xa, Xb)
    # no callback => fully asynchronous device path; same final answer
    r_async = sa.optimize.minimize(sa.factory.sphere, bounds, method="cpso", options=dict(opts, backend="hip", rng="philox"))
    assert r_async.fun == r_ref.fun and np.array_equal(r_async.x, r_ref.x) and r_async.nit == r_ref.nit


@pytest.mark.parametrize("objective,n,P,maxiter,shrink", [("sphere", 16, 256, 120, None), ("ackley", 32, 1024, 150, None),
                                                          ("rosenbrock", 130, 512, 80, "Shrink"), ("sphere", 8, 4000, 60, None),
                                                          ("sphere", 600, 48, 110, "Shrink"), ("rosenbrock", 300, 512, 60, None),
                                                          ("sphere", 40, 130, 170, "Shrink")])
def test_cpso_graph_path_equals_stepping_and_oracle(sa, objective, n, P, maxiter, shrink):
    """Inside a replayed graph (no callback, no history) a decided restart is carried out by the NEXT generation kernel,
    which re-seeds the selected rows instead of loading them (sx_pso_args.pending_restart); the apply kernel runs once
    at the end of a replay.  The run is the generation-by-generation one (history: every restart applied by its own
    kernel) bit for bit, and -- for the +,-,* objectives -- the oracle's, restarts included."""
    opts = {"maxiter": maxiter, "popsize": P, "seed": 21, "updating": "deferred", "backend": "hip", "rng": "philox",
            "constraints": shrink}
    bounds = [[-5.12, 5.12]] * n
    fun = getattr(sa.factory, objective)
    graph = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts))
    step = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts, return_all=True))
    assert graph.fun == step.fun and np.array_equal(graph.x, step.x) and (graph.nit, graph.status) == (step.nit, step.status)
    if objective != "ackley":  # restarts do fire in these runs (the oracle counts them)
        ref = oracle.minimize(objective, bounds, method="cpso", rng="philox",
                              options={k: v for k, v in opts.items() if k not in ("backend", "rng")})
        assert len(ref["_restarts"]) > 3, len(ref["_restarts"])
        assert ref.fun == graph.fun and np.array_equal(ref.x, graph.x) and ref.nit == graph.nit


@pytest.mark.parametrize("objective,n,P,maxiter,shrink", [("sphere", 64, 1000, 150, None), ("rosenbrock", 128, 512, 120, "Shrink"),
                                                          ("sphere", 256, 700, 130, None), ("ackley", 256, 4096, 80, None),
                                                          ("sphere", 256, 130, 170, "Shrink")])
def test_cpso_graph_two_launches_per_generation_all_three_forms_agree(sa, objective, n, P, maxiter, shrink, monkeypatch):
    """Whole-batch rows (n = 64 / 128 / 256) inside a graph: the generation kernel leaves the radius against the OLD best and
    cpso_post_kernel (best / termination + restart decision + selection) follows -- two launches per generation.  The run
    is, bit for bit, (a) the four-launch graph (SX_CPSO_FUSED_RADIUS=0: best / termination, radius pass over X, selection),
    (b) the same graph with every generation forced through the rare branch (SX_CPSO_FORCE_EXACT=1: the post kernel's own
    radii against the new best for the rows that can hold the maximum -- the bound needs it in ~0.3 % of generations; =2: for
    all rows, what it falls back to when the candidates do not fit its list) and (c) the generation-by-generation run; restarts fire
    (counted by the oracle for the +,-,* objectives, which must agree as well)."""
    opts = {"maxiter": maxiter, "popsize": P, "seed": 5 + n, "updating": "deferred", "backend": "hip", "rng": "philox",
            "constraints": shrink}
    bounds = [[-5.12, 5.12]] * n
    fun = getattr(sa.factory, objective)
    fused = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts))
    monkeypatch.setenv("SX_CPSO_FORCE_EXACT", "1")
    forced = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts))
    monkeypatch.setenv("SX_CPSO_FORCE_EXACT", "2")
    forced_all = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts))
    monkeypatch.delenv("SX_CPSO_FORCE_EXACT")
    monkeypatch.setenv("SX_CPSO_FUSED_RADIUS", "0")
    four = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts))
    monkeypatch.delenv("SX_CPSO_FUSED_RADIUS")
    step = sa.optimize.minimize(fun, bounds, method="cpso", options=dict(opts, return_all=True))
    for other in (forced, forced_all, four, step):
        assert fused.fun == other.fun and np.array_equal(fused.x, other.x)
        assert (fused.nit, fused.status) == (other.nit, other.status)
    if objective != "ackley":
        ref = oracle.minimize(objective, bounds, method="cpso", rng="philox",
                              options={k: v for k, v in opts.items() if k not in ("backend", "rng")})
        assert len(ref["_restarts"]) > 3, len(ref["_restarts"])
        assert ref.fun == fused.fun and np.array_equal(ref.x, fused.x) and ref.nit == fused.nit


@pytest.mark.parametrize("objective,n,P,maxiter,ftol", [
    ("sphere", 64, 40, 70, -1.0), ("rosenbrock", 128, 33, 45, -1.0), ("sphere", 256, 70, 301, -1.0),
    ("ackley", 256, 2048, 90, -1.0), ("sphere", 64, 512, 4000, 1e-3), ("sphere", 128, 300, 4000, 30.0),
    ("rastrigin", 256, 16384, 40, -1.0)])
def test_pso_chained_kernel_equals_two_kernel_path_and_oracle(sa, objective, n, P, maxiter, ftol, monkeypatch):
    """Plain PSO on whole-batch rows CAN run one kernel per generation (opt-in SX_PSO_CHAIN=1; csrc/sx_pso.hip CHAIN: best / termination in the next
    launch's prologue, gbest read from per-workgroup best-row copies; replayed 32-generation graphs + eager tail + the
    finalise-only looks).  Same run, bit for bit, as the generation + select_finalize pair (SX_PSO_CHAIN=0) and as the
    oracle: fun, x, nit, status -- including stops on ftol (status 0 / 1 settled from the two resident best rows) and
    run lengths that are no multiple of the graph length."""
    from stochopy_amd.optimize import _cpso

    opts = {"maxiter": maxiter, "popsize": P, "seed": 31 + n, "updating": "deferred", "ftol": ftol, "xtol": 1e-9}
    bounds = [[-3.0, 2.0]] * n
    chained = []
    orig = _cpso._PsoRun._setup

    def spy(self):
        orig(self)
        chained.append(self.chain)

    monkeypatch.setattr(_cpso._PsoRun, "_setup", spy)
    monkeypatch.setenv("SX_PSO_CHAIN", "1")  # (opt-in: measured slower than the two-kernel path, see _cpso.py)
    got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="pso", options=dict(opts, backend="hip", rng="philox"))
    monkeypatch.setenv("SX_PSO_CHAIN", "0")
    two = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method="pso", options=dict(opts, backend="hip", rng="philox"))
    assert chained == [True, False]
    assert (got.fun, got.nit, got.nfev, got.status, got.message) == (two.fun, two.nit, two.nfev, two.status, two.message)
    assert np.array_equal(got.x, two.x)
    if P <= 4096:
        ref = oracle.minimize(objective, bounds, method="pso", options=dict(opts), rng="philox")
        assert (got.nit, got.status) == (ref.nit, ref.status)
        if objective in ("sphere", "rosenbrock"):
            assert got.fun == ref.fun and np.array_equal(got.x, ref.x)
        else:
            assert np.isclose(got.fun, ref.fun, rtol=1e-6)
    if ftol > 0:
        assert got.status in (0, 1) and got.nit < maxiter


def _sort_key_np(f):
    b = np.asarray(f, dtype=np.float64).view(np.uint64)
    return np.where(b >> np.uint64(63), ~b, b | np.uint64(0x8000000000000000))


@pytest.mark.parametrize("P", [8192, 12000, 16384])
@pytest.mark.parametrize("kind", ["spread", "converged", "outliers", "ties", "two_values", "reseeded"])
def test_restart_selection_kernel_vs_numpy(sa, P, kind):
    """sx_pso_restart_select at the swarm sizes where it first cuts the keys down to a sample-guided window (round 3:
    256 samples ranked, the target's sample rank +- 24, keys above counted, keys inside packed into LDS) before the radix
    descent: nw and the threshold key against numpy for fitness sets that stress the window -- spread values, a converged
    swarm (common leading bits), a few huge outliers, many exact ties (the window overflows / the target sits in a tie),
    two distinct values only, and freshly re-seeded particles at 1e30 -- across generation numbers that move nw from
    P - 1 down to 1."""
    import ctypes as C

    from stochopy_amd import _device, _lib

    ctx = _device.Context()
    L, t, p = ctx.L, _device.torch(), _device.ptr
    rs = np.random.RandomState(P + len(kind))
    n, maxiter, delta, gamma = 8, 1000, 1.0e-2, 1.0
    if kind == "spread":
        fit = rs.uniform(0.0, 100.0, P)
    elif kind == "converged":
        fit = 3.7 + 1.0e-9 * rs.rand(P)
    elif kind == "outliers":
        fit = 1.0e-3 * rs.rand(P)
        fit[rs.choice(P, 7, replace=False)] = 10.0 ** rs.uniform(3, 200, 7)
    elif kind == "ties":
        fit = np.round(rs.rand(P) * 20.0) / 20.0
    elif kind == "two_values":
        fit = np.where(rs.rand(P) < 0.3, 1.0, 2.0)
    else:
        fit = 0.5 + 1.0e-6 * rs.rand(P)
        fit[rs.choice(P, P // 3, replace=False)] = 1.0e30
    with t.cuda.stream(ctx.stream):
        g = ctx.L.sx_num_partials(P, n)
        d = {k: ctx.zeros((P, n)) for k in ("X", "V", "pbest")}
        d_fit = ctx.upload(fit)
        d_gbest, d_pf, d_pi = ctx.zeros((n,)), ctx.zeros((g,)), ctx.zeros((g,), dtype=t.int64)
        d_partr = ctx.upload(np.full(g, 1.0e-6))  # radius far below delta: a restart is due
        out = ctx.zeros((3,), dtype=t.int64)
        for it in (40, 300, 480, 520, 560, 640, 760, 900):
            st = _lib.SxState(it=it, gbidx=0, gfit=0.0, dx=0.0, status=_lib.SX_STATUS_NONE, done=0)
            d_state = ctx.upload(np.frombuffer(bytes(st), dtype=np.int64).copy())
            a = _lib.SxPsoArgs()
            a.X, a.V, a.pbest, a.pbestfit, a.gbest = (x.data_ptr() for x in (d["X"], d["V"], d["pbest"], d_fit, d_gbest))
            a.state, a.part_f, a.part_i = d_state.data_ptr(), d_pf.data_ptr(), d_pi.data_ptr()
            a.P, a.ld, a.row0, a.n, a.fun_id, a.rng, a.maxiter = P, n, 0, n, 0, _lib.SX_RNG_PHILOX, maxiter
            _lib.check(L.sx_pso_restart_select(C.byref(a), p(d_partr), delta, gamma, p(out), ctx.stream_ptr),
                       "sx_pso_restart_select")
            ctx.sync()
            got = out.cpu().numpy().view(np.uint64)
            nw = int((P - 1.0) / (1.0 + np.exp(1.0 / 0.09 * (it / maxiter - gamma + 0.5))))
            assert int(got[0]) == max(nw, 0), (it, got[0], nw)
            if nw > 0:
                want = np.sort(_sort_key_np(fit))[::-1][nw - 1]
                assert got[1] == want, (kind, P, it, nw, hex(int(got[1])), hex(int(want)))
