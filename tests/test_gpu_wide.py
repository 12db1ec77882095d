"""GPU parity for WIDE rows (n > 4096; csrc/sx_wide.hip: one workgroup per individual, the summation plan in device
memory, the row resident in LDS up to 8794 elements -- two workgroups per CU -- and streamed through a 4096-element stage above).  The reference
has no dimension limit (de/_de.py:208-218; vdcma/_vdcma.py:144-458 exists for long vectors): objectives, DE, PSO, CPSO
and the unfused / sharded paths are compared with the oracle BIT FOR BIT (+, -, * objectives), VD-CMA within 1e-6 and
against a vector captured from the reference at n = 8192 (tests/golden/vdcma_wide.json)."""
import os

import numpy as np
import pytest

import oracle
from oracle.objectives import OBJECTIVES
from conftest import GOLDEN, load_golden, unhex

pytestmark = pytest.mark.gpu

EXACT = {"rosenbrock", "sphere"}


@pytest.fixture(scope="module")
def sa():
    import stochopy_amd

    return stochopy_amd


# 4097: first wide length (a tail term); 4104 / 5000: tails of 0 / 7 for Rosenbrock; 16384: a power of two;
# 18400, 20001: several chunks, a tail (sx_eval always streams); 65536: the verdict's
# "at least"; 100003: beyond it
# 2048 / 2049: the last row of the wavefront-per-row kernels and the first of these (kWideFrom, round 5; 2560 / 2561 until the
# wide kernels' second pass); 3000, 4096: rows the
# wavefront-per-row kernels served until then
# 8193 / 8199 / 8200: a second 8192-term piece of 1 / 7 / 8 terms (no leaf block / one block); 12295: a last piece of 4103
# terms (33 leaves + a tail of 7); 16383 / 24575: last pieces of 8191 / 8190 terms -- 65 leaves in 7 levels, the most a piece
# can have (the per-piece finish reads the 65th where it lies); 262144: the limit
@pytest.mark.parametrize("n", [2048, 2049, 2056, 2560, 2561, 3000, 4096, 4097, 4104, 5000, 8192, 8193, 8199, 8200, 12295, 16383, 16384, 18400,
                               20001, 24575, 65536, 100003, 262144])
@pytest.mark.parametrize("name", sorted(OBJECTIVES))
def test_wide_objectives_vs_oracle(sa, name, n):
    rs = np.random.RandomState(n % 1000 + 3)
    X = rs.uniform(-5.12, 5.12, (11, n))
    got = getattr(sa.factory, name)(X)
    ref = OBJECTIVES[name](X)
    if name in EXACT:
        assert np.array_equal(got, ref)
    else:
        assert np.allclose(got, ref, rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("n", [4097, 16384, 70000])
def test_wide_eval_affine_clip_and_penalty(sa, n):
    """CMA-ES-family evaluation of wide rows (VD-CMA's candidates): un-standardisation (cmaes/_cmaes.py:171), clipping and
    the weighted squared excess of Penalize (cmaes/_constraints.py:29-31, :79)."""
    import torch
    from stochopy_amd import _device, _lib
    from stochopy_amd.optimize import _common

    rs = np.random.RandomState(n % 97)
    P = 9
    X = rs.uniform(-1.4, 1.4, (P, n))
    xm, xstd, v = rs.uniform(-1, 1, n), rs.uniform(0.5, 2.0, n), rs.uniform(0.0, 2.0, n)
    ctx = _device.Context()
    d = lambda a: torch.as_tensor(a, device=ctx.device)
    Xd, f, pen = d(X), ctx.empty((P,)), ctx.empty((P,))
    fid = _lib.FUN_IDS["rosenbrock"]
    _common.evaluate_rows(ctx, fid, Xd, n, f, xm=d(xm), xstd=d(xstd))
    ctx.sync()
    assert np.array_equal(f.cpu().numpy(), OBJECTIVES["rosenbrock"](X * xstd + xm))
    _common.evaluate_rows(ctx, fid, Xd, n, f, xm=d(xm), xstd=d(xstd), clip=True)
    ctx.sync()
    Xc = np.clip(X, -1.0, 1.0)
    assert np.array_equal(f.cpu().numpy(), OBJECTIVES["rosenbrock"](Xc * xstd + xm))
    _common.penalty_rows(ctx, fid, Xd, n, d(xm), d(xstd), d(v), f, pen)
    ctx.sync()
    assert np.allclose(pen.cpu().numpy(), (((Xc - X) ** 2) * v).sum(axis=1), rtol=1e-12)


def _trace_pair(sa, objective, bounds, method, opts):
    t_ref, t_got = [], []
    r_ref = oracle.minimize(objective, bounds, method=method, options=dict(opts), rng="philox",
                            callback=lambda X, r: t_ref.append((r.fun, X.copy())))
    r_got = sa.optimize.minimize(getattr(sa.factory, objective), bounds, method=method,
                                 options=dict(opts, backend="hip", rng="philox"),
                                 callback=lambda X, r: t_got.append((r.fun, X.copy())))
    return r_ref, r_got, t_ref, t_got


def _same_run(r_ref, r_got, t_ref, t_got):
    assert len(t_ref) == len(t_got)
    for g, ((fa, Xa), (fb, Xb)) in enumerate(zip(t_ref, t_got)):
        assert fa == fb, g
        assert np.array_equal(Xa, Xb), g
    assert np.array_equal(r_ref.x, r_got.x) and (r_ref.nit, r_ref.status) == (r_got.nit, r_got.status)
    assert r_ref.fun == r_got.fun


@pytest.mark.parametrize("strategy,constraints", [("best1bin", None), ("rand1bin", "Random"), ("rand2bin", None),
                                                  ("best2bin", "Random")])
# (8794 / 8795: the longest resident row and the first streamed one)
@pytest.mark.parametrize("n,P", [(2048, 24), (2049, 24), (2560, 24), (2561, 24), (3000, 20), (4097, 24), (8192, 16), (8794, 12),
                                 (8795, 12), (16384, 12), (20001, 10)])
def test_wide_de_philox_matches_oracle(sa, strategy, constraints, n, P):
    """DE with in-kernel draws, whole populations of every generation, resident (<= 8794 elements) and streamed rows."""
    opts = {"maxiter": 6, "popsize": P, "seed": 77 + n, "strategy": strategy, "constraints": constraints,
            "mutation": 0.7, "recombination": 0.6, "updating": "deferred"}
    _same_run(*_trace_pair(sa, "rosenbrock", [[-2.0, 2.0]] * n, "de", opts))


def test_wide_de_graph_equals_stepwise_and_oracle(sa):
    """No callback: replayed graphs of (wide generation kernel, best / termination) -- the same run."""
    n, P = 8192, 64
    bounds = [[-5.12, 5.12]] * n
    o = {"maxiter": 120, "popsize": P, "seed": 5, "updating": "deferred", "backend": "hip", "rng": "philox",
         "ftol": -1.0, "xtol": 0.0}
    a = sa.optimize.minimize(sa.factory.sphere, bounds, method="de", options=dict(o))
    b = sa.optimize.minimize(sa.factory.sphere, bounds, method="de", options=dict(o), callback=lambda X, r: None)
    assert a.nit == b.nit == 120 and a.fun == b.fun and np.array_equal(a.x, b.x)
    ref = oracle.minimize("sphere", bounds, method="de", options={"maxiter": 120, "popsize": P, "seed": 5, "ftol": -1.0, "xtol": 0.0, "updating": "deferred"},
                          rng="philox")
    assert ref.fun == a.fun and np.array_equal(ref.x, a.x)


def test_wide_de_numpy_legacy_matches_oracle(sa):
    """The reference's own random stream (host draws uploaded per generation) through the wide kernel."""
    n, P = 4100, 10
    opts = {"maxiter": 5, "popsize": P, "seed": 3, "strategy": "rand1bin", "constraints": "Random", "updating": "deferred"}
    bounds = [[-3.0, 3.0]] * n
    ref = oracle.minimize("rosenbrock", bounds, method="de", options=dict(opts), rng="numpy-legacy")
    got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(opts, backend="hip", rng="numpy-legacy"))
    assert ref.fun == got.fun and np.array_equal(ref.x, got.x) and ref.nit == got.nit


def test_wide_default_call_defers_with_a_warning(sa):
    """updating="immediate" (the reference's default) is an ordered sweep that keeps a row in one workgroup's LDS: rows of
    more than 4096 elements run deferred, as with a parallel backend of the reference (de/_de.py:142-145), and say so."""
    n = 4500
    with pytest.warns(RuntimeWarning, match="deferred"):
        r = sa.optimize.minimize(sa.factory.sphere, [[-1.0, 1.0]] * n, method="de",
                                 options={"maxiter": 3, "popsize": 8, "seed": 0, "rng": "philox"})
    ref = oracle.minimize("sphere", [[-1.0, 1.0]] * n, method="de", options={"maxiter": 3, "popsize": 8, "seed": 0, "updating": "deferred"},
                          rng="philox")
    assert r.fun == ref.fun and np.array_equal(r.x, ref.x)


@pytest.mark.parametrize("constraints", [None, "Shrink"])
@pytest.mark.parametrize("n,P", [(2048, 20), (2049, 20), (2560, 20), (2561, 20), (3500, 16), (4097, 20), (8192, 16), (8794, 10),
                                 (8795, 10), (20001, 9)])
def test_wide_pso_philox_matches_oracle(sa, constraints, n, P):
    opts = {"maxiter": 6, "popsize": P, "seed": 11 + n, "constraints": constraints, "updating": "deferred"}
    _same_run(*_trace_pair(sa, "rosenbrock", [[-2.0, 2.0]] * n, "pso", opts))


@pytest.mark.parametrize("n,P", [(4097, 24), (19000, 12)])
def test_wide_cpso_restarts_match_oracle(sa, n, P):
    """Competitive restarts (cpso/_cpso.py:405-426) on wide rows, stepwise (callback: radius / select / apply kernels) and
    through replayed graphs (re-seeding inside the next generation kernel): a large competitivity on a tight box makes
    the swarm restart within a few generations."""
    opts = {"maxiter": 40, "popsize": P, "seed": 21, "competitivity": 1.9, "constraints": "Shrink", "updating": "deferred",
            "ftol": -1.0, "xtol": 0.0}
    bounds = [[-0.01, 0.01]] * n
    r_ref, r_got, t_ref, t_got = _trace_pair(sa, "sphere", bounds, "cpso", opts)
    _same_run(r_ref, r_got, t_ref, t_got)
    g = sa.optimize.minimize(sa.factory.sphere, bounds, method="cpso", options=dict(opts, backend="hip", rng="philox"))
    assert g.fun == r_ref.fun and np.array_equal(g.x, r_ref.x) and g.nit == r_ref.nit


def test_wide_rows_around_a_caller_supplied_objective(sa):
    """factory.batched (propose / move -> caller's device objective -> select): the same run as the fused wide kernels."""
    import torch

    n, P = 5000, 12
    bounds = [[-2.0, 2.0]] * n

    @sa.factory.batched
    def sphere_dev(X):
        from stochopy_amd import _device, _lib

        ctx = _device.Context()
        return _device.evaluate(ctx, _lib.FUN_IDS["sphere"], X, n)

    for method, extra in (("de", {"strategy": "rand1bin", "constraints": "Random"}), ("cpso", {"constraints": "Shrink", "competitivity": 1.0})):
        o = dict({"maxiter": 6, "popsize": P, "seed": 4, "updating": "deferred", "backend": "hip", "rng": "philox"}, **extra)
        a = sa.optimize.minimize(sa.factory.sphere, bounds, method=method, options=dict(o))
        b = sa.optimize.minimize(sphere_dev, bounds, method=method, options=dict(o))
        assert a.fun == b.fun and np.array_equal(a.x, b.x) and a.nit == b.nit


def test_wide_rows_sharded_one_rank_group(sa, monkeypatch):
    """The sharded path (records, one all-gather, sx_gather_finalize) with wide rows: a 1-rank process group is forced
    through it and must reproduce the single-GPU run."""
    import torch.distributed as dist

    n, P = 6000, 16
    bounds = [[-2.0, 2.0]] * n
    o = {"maxiter": 6, "popsize": P, "seed": 8, "updating": "deferred", "backend": "hip", "rng": "philox"}
    a = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o))
    p = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="pso", options=dict(o))
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29653")
        dist.init_process_group("gloo", rank=0, world_size=1)
        created = True
    try:
        monkeypatch.setenv("SX_FORCE_SHARDED", "1")
        b = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="de", options=dict(o))
        q = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="pso", options=dict(o))
    finally:
        if created:
            dist.destroy_process_group()
    assert a.fun == b.fun and np.array_equal(a.x, b.x) and a.nit == b.nit
    assert p.fun == q.fun and np.array_equal(p.x, q.x) and p.nit == q.nit


# (3000: VD-CMA's narrow model kernels -- n <= 4096 -- with the objective through the one-workgroup-per-row kernel)
# (2100, 1100): more candidates than a round of workgroups
@pytest.mark.parametrize("n,P,maxiter", [(2100, 12, 10), (2100, 1100, 4), (3000, 12, 10), (4097, 12, 12), (8192, 20, 10),
                                         (16384, 16, 8), (70000, 8, 5)])
def test_wide_vdcma_device_loop_matches_oracle(sa, n, P, maxiter, monkeypatch):
    """VD-CMA's device-resident loop with the wide model-update kernel (csrc/sx_vd_loop.hip vd_update_wide_kernel) and the
    wide objective, against the oracle's numpy loop: best-f of every generation within 1e-6; the host-driven loop
    (numpy model update around the same sampling / objective kernels) must agree too."""
    opts = {"maxiter": maxiter, "popsize": P, "seed": 5, "sigma": 0.3}
    bounds = [[-3.0, 3.0]] * n
    t_ref, t_got, t_host = [], [], []
    ref = oracle.minimize("rosenbrock", bounds, method="vdcma", options=dict(opts), rng="philox",
                          callback=lambda X, r: t_ref.append(r.fun))
    got = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="vdcma", options=dict(opts, backend="hip", rng="philox"),
                               callback=lambda X, r: t_got.append(r.fun))
    assert np.allclose(t_got, t_ref, rtol=1e-6) and (got.nit, got.status) == (ref.nit, ref.status)
    assert np.allclose(got.x, ref.x, rtol=1e-5, atol=1e-6)
    quiet = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="vdcma", options=dict(opts, backend="hip", rng="philox"))
    assert quiet.fun == got.fun and np.array_equal(quiet.x, got.x)
    monkeypatch.setenv("SX_CMA_LOOP", "host")
    host = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="vdcma", options=dict(opts, backend="hip", rng="philox"),
                                callback=lambda X, r: t_host.append(r.fun))
    assert np.allclose(t_host, t_ref, rtol=1e-6) and (host.nit, host.status) == (ref.nit, ref.status)


_CHAIN_AB = r"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.environ["SX_REPO"])
import stochopy_amd as sa
out = []
for n, P, maxiter, extra in [(4097, 12, 12, {}), (8192, 64, 9, {}), (16384, 40, 6, {"constraints": "Penalize"}),
                             (5000, 16, 300, {"ftol": 1.0e8}), (70000, 8, 5, {}), (100003, 6, 4, {})]:
    opts = dict({"maxiter": maxiter, "popsize": P, "seed": 11, "sigma": 0.3, "backend": "hip", "rng": "philox"}, **extra)
    trace = []
    r = sa.optimize.minimize(sa.factory.rosenbrock, [[-3.0, 3.0]] * n, method="vdcma", options=opts,
                             callback=lambda X, r: trace.append(float(r.fun).hex()))
    out.append([trace, float(r.fun).hex(), r.x.tobytes().hex()[:4096], int(r.nit), int(r.nfev), int(r.status)])
print("RESULT" + json.dumps(out))
"""


def test_wide_vdcma_chain_in_one_launch_is_the_chain_of_launches():
    """The wide model update as ONE launch with grid-wide barriers (csrc/sx_vd_loop.hip vw_chain_kernel, the default) against
    the same phases as one launch each (SX_VD_CHAIN=0, read once per process: two child processes): every generation's
    best-f, the result and the counters bit for bit -- ordinary runs, Penalize, a run that stops on ftol, n up to 100 003."""
    import json
    import subprocess
    import sys

    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for chain in ("1", "0"):
        env = dict(os.environ, SX_VD_CHAIN=chain, SX_REPO=repo)
        p = subprocess.run([sys.executable, "-c", _CHAIN_AB], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stderr[-2000:]
        got[chain] = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("RESULT")][0][6:])
    assert got["1"] == got["0"]
    assert got["1"][3][5] == 1 and got["1"][3][3] < 300  # (the ftol run stopped early)


def test_wide_vdcma_matches_the_reference_at_n_8192(sa):
    """A run captured from the reference itself (tests/golden/make_golden.py, numpy-legacy draws, n = 8192 -- the size
    VD-CMA exists for, vdcma/_vdcma.py:144-458): best-f of every generation within 1e-6, the result within 1e-5."""
    case = load_golden("vdcma_wide.json")["cases"][0]
    trace = []
    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), [case["bounds"][0]] * case["ndim"], method="vdcma",
                               options=opts, callback=lambda X, r: trace.append(float(r.fun)))
    want = unhex(case["fun_trace"])
    ref = case["result"]
    assert len(trace) == len(want) and np.allclose(trace, want, rtol=1e-6, atol=1e-300)
    assert (res.nit, res.status) == (ref["nit"], ref["status"])
    assert np.isclose(res.fun, unhex(ref["fun"]), rtol=1e-6, atol=0)
    arrays = np.load(os.path.join(GOLDEN, "vdcma_wide.npz"))
    assert np.allclose(res.x, arrays[case["tag"] + "__x"], rtol=1e-5, atol=1e-7)


_REF_RUNS = [c for c in load_golden("vdcma_wide.json")["cases"] if c["method"] != "vdcma"]


@pytest.mark.parametrize("case", _REF_RUNS, ids=lambda c: c["tag"])
def test_wide_runs_captured_from_the_reference(sa, case):
    """DE / PSO / CPSO at n = 4097 ... 9000 with the reference's own random stream, seed for seed the reference's run
    (tests/golden/make_golden.py vdcma_wide): best-f of every generation, the result and the last population, bit for bit."""
    trace, pops = [], []
    opts = dict(case["options"], backend="hip", rng="numpy-legacy")
    res = sa.optimize.minimize(getattr(sa.factory, case["objective"]), [case["bounds"][0]] * case["ndim"],
                               method=case["method"], options=opts,
                               callback=lambda X, r: (trace.append(float(r.fun)), pops.append(X.copy())))
    ref = case["result"]
    assert np.array_equal(np.array(trace), unhex(case["fun_trace"]))
    assert (res.nit, res.nfev, res.status) == (ref["nit"], ref["nfev"], ref["status"]) and float(res.fun).hex() == ref["fun"]
    arrays = np.load(os.path.join(GOLDEN, "vdcma_wide.npz"))
    assert np.array_equal(res.x, arrays[case["tag"] + "__x"])
    assert np.array_equal(pops[-1], arrays[case["tag"] + "__pop_last"])


def test_wide_objective_values_from_the_reference(sa):
    arrays = np.load(os.path.join(GOLDEN, "vdcma_wide.npz"))
    for key, want in load_golden("vdcma_wide.json")["objective_kat"].items():
        name, n = key.rsplit("_", 1)
        got = getattr(sa.factory, name)(arrays["kat_X_" + n])
        if name in EXACT:
            assert np.array_equal(got, unhex(want)), key
        else:
            assert np.allclose(got, unhex(want), rtol=1e-13, atol=1e-13), key


def test_dimension_limits_are_loud(sa):
    """What is left of the old n <= 4096 limit: rows beyond 262 144 elements, and full CMA-ES (n x n covariance)."""
    from stochopy_amd._lib import HipLibraryError, WIDE_DIM

    with pytest.raises((HipLibraryError, ValueError)):
        sa.optimize.minimize(sa.factory.sphere, [[-1.0, 1.0]] * (WIDE_DIM + 1), method="de",
                             options={"maxiter": 2, "popsize": 4, "seed": 0, "updating": "deferred", "rng": "philox"})
    with pytest.raises(ValueError, match="vdcma"):
        sa.optimize.minimize(sa.factory.sphere, [[-1.0, 1.0]] * 4097, method="cmaes", options={"maxiter": 2, "popsize": 8, "seed": 0})
