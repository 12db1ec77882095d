"""CPU-only tests: host RNG replica vs numpy, the C-ABI surface, host-side API logic.

No GPU: these load the shared library (hipcc cross-compiled it) and call only host entry points.
"""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, load_golden, unhex


@pytest.fixture(scope="module")
def lib():
    from stochopy_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__

        __graft_entry__.build()
    return _lib.lib()


def test_library_exports_every_declared_symbol(lib):
    """Every function include/stochopy_hip.h declares must resolve in the .so (and be bound in _lib.PROTOTYPES)."""
    from stochopy_amd import _lib

    header = open(os.path.join(ROOT, "include", "stochopy_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(sx_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} not exported"
        assert name in _lib.PROTOTYPES, f"{name} has no ctypes prototype"
    assert set(_lib.PROTOTYPES) <= declared | {"sx_abi_version"}
    assert lib.sx_abi_version() == 1


def test_struct_layouts_match_the_header(lib):
    from stochopy_amd import _lib

    assert C.sizeof(_lib.SxState) == 64 == lib.sx_struct_size(0)
    assert C.sizeof(_lib.SxDeArgs) == lib.sx_struct_size(1)
    assert C.sizeof(_lib.SxPsoArgs) == lib.sx_struct_size(2)
    assert C.sizeof(_lib.SxXchgArgs) == lib.sx_struct_size(3)
    assert C.sizeof(_lib.SxCmaState) == lib.sx_struct_size(4)
    assert C.sizeof(_lib.SxCmaArgs) == lib.sx_struct_size(5)
    assert C.sizeof(_lib.SxVdArgs) == lib.sx_struct_size(6)
    # field offsets of the scalars that follow the pointer block
    assert _lib.SxDeArgs.P.offset == 14 * 8 and _lib.SxDeArgs.key0.offset == C.sizeof(_lib.SxDeArgs) - 8
    assert _lib.SxPsoArgs.P.offset == 14 * 8 and _lib.SxPsoArgs.key0.offset == C.sizeof(_lib.SxPsoArgs) - 8


class Stream:
    def __init__(self, seed):
        from stochopy_amd import _rng

        state = np.random.get_state()
        self.s = _rng.LegacyHostStream(seed)
        np.random.set_state(state)


def test_rng_stream_golden(lib):
    """tests/golden/rng_stream.json: the call sequence captured from numpy's legacy global stream."""
    g = load_golden("rng_stream.json")
    s = Stream(g["seed"]).s
    lo, hi = np.array([-1.0, 0.0, 2.5]), np.array([1.0, 10.0, 2.75])
    for call in g["calls"]:
        name = call["call"]
        if name == "rand(3,4)":
            got = s.random((3, 4)).ravel()
        elif name.startswith("permutation(delete"):
            p = s.permutation(7)
            got = np.delete(np.arange(8), 3)[p]
        elif name == "permutation(8)":
            got = s.permutation(8)
        elif name == "randint(128,size=10)":
            got = s.randint(128, 10)
        elif name == "randint(2)":
            got = s.randint(2, 1)
        elif name == "randint(100,size=7)":
            got = s.randint(100, 7)
        elif name in ("randn(3)",):
            got = s.randn((3,))
        elif name.startswith("uniform(lo,hi,(4,3))"):
            got = s.uniform_rows(lo, hi, 4).ravel()
        elif name == "uniform(-1,1,5)":
            got = s.uniform(-1.0, 1.0, 5)
        elif name == "uniform(size=(2,3))":
            got = s.random((2, 3)).ravel()
        elif name == "uniform(0.25,0.75)":
            got = s.uniform(0.25, 0.75, 1)
        elif name == "normal(0,1,4)":
            got = s.randn((4,))
        elif name.startswith("randint(2**32"):
            got = s.randint(4294967296, 4)
        else:
            raise AssertionError(name)
        want = unhex(call["f64"]) if "f64" in call else np.array(call["i64"])
        assert np.array_equal(np.asarray(got).ravel(), np.asarray(want).ravel()), name
    for entry in g["long"]:
        s = Stream(entry["seed"]).s
        import hashlib

        d = s.random((1000,))
        assert hashlib.sha256(d.tobytes()).hexdigest() == entry["rand1000_sha"]
        z = s.randn((1001,))
        assert hashlib.sha256(z.tobytes()).hexdigest() == entry["randn1001_sha"]
        assert [int(v) for v in s.permutation(257)] == entry["permutation257"]
        r = s.randint(1000, 500)
        assert int(r.sum()) == entry["randint1000x500_sum"]


@pytest.mark.parametrize("seed", [0, 1, 42, 2**32 - 1])
def test_rng_matches_numpy_live(lib, seed):
    """Same draws as numpy.random.RandomState, incl. the DE donor permutations and LHS."""
    s = Stream(seed).s
    rs = np.random.RandomState(seed)
    assert np.array_equal(s.random((5, 7)), rs.rand(5, 7))
    P, k = 37, 5
    don = s.de_donors(P, k)
    ref = np.transpose([rs.permutation(np.delete(np.arange(P), i)) for i in range(P)])[:k]
    assert np.array_equal(don, ref)
    assert np.array_equal(s.randint(13, P), rs.randint(13, size=P))
    lo, hi = np.linspace(-3, -1, 6), np.linspace(1, 4, 6)
    assert np.array_equal(s.uniform_rows(lo, hi, 9), rs.uniform(lo, hi, (9, 6)))
    assert np.array_equal(s.randn((4, 5)), rs.randn(4, 5))  # leaves a cached gaussian
    assert np.array_equal(s.randn((3,)), rs.randn(3))


@pytest.mark.parametrize("P,k,seed,skip", [(256, 2, 1, 0), (300, 5, 2, 7), (1000, 3, 3, 1), (2048, 2, 5, 11), (4096, 2, 0, 5),
                                           (4097, 4, 9, 3), (257, 3, 4, 623)])
@pytest.mark.parametrize("scalar", [False, True, "wide-index"])
def test_large_population_donors_match_numpy(lib, P, k, seed, skip, scalar):
    """de/_de.py:304-311 at population sizes that take the wide form of the donor draws (csrc/sx_mt19937.cpp: masked
    rejection 64 words at a time, the permutation's first k entries by walking the swaps backwards -- no array is
    shuffled): the same donors as numpy's legacy permutation() of every individual's index list, and the stream left at
    the same word (the draws behind it agree).  Both forms (SX_MT_SCALAR=1: the plain replay) in a fresh process each,
    the CPU check is made once per process.  "wide-index" (SX_MT_WIDE_INDEX=1): the 32-bit-index instantiation
    of the wide form -- what populations of more than 65535 individuals (BASELINE config 5 with rng="numpy-legacy") run -- at
    these small sizes, where numpy itself can still be asked."""
    import subprocess
    import sys

    code = f"""
import sys, numpy as np
sys.path.insert(0, {ROOT!r})
from stochopy_amd import _rng
s = _rng.LegacyHostStream({seed})
rs = np.random.RandomState({seed})
if {skip}:
    assert np.array_equal(s.random({skip}), rs.rand({skip}))
got = s.de_donors({P}, {k})
ref = np.empty(({k}, {P}), dtype=np.int32)
idx = np.arange({P})
for i in range({P}):
    ref[:, i] = rs.permutation(np.delete(idx, i))[:{k}]
assert np.array_equal(got, ref), "donors differ"
assert np.array_equal(s.randint(1000, 8), rs.randint(1000, size=8)), "stream position differs"
assert np.array_equal(s.random((3, 5)), rs.rand(3, 5))
print("ok")
"""
    env = dict(os.environ)
    env.pop("SX_MT_SCALAR", None)
    env.pop("SX_MT_WIDE_INDEX", None)
    if scalar == "wide-index":
        env["SX_MT_WIDE_INDEX"] = "1"
    elif scalar:
        env["SX_MT_SCALAR"] = "1"
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("P,k,n,bounded", [(9, 3, 1, False), (37, 5, 13, True), (130, 2, 64, True), (40, 4, 5, False)])
def test_async_de_draws_match_numpy(lib, P, k, n, bounded):
    """updating="immediate": per individual donor permutation, randint(ndim), Random's uniform(lower, upper, n)
    (de/_de.py:376-382), against numpy itself and against the oracle's stream."""
    from oracle.streams import LegacyStream

    s = Stream(5).s
    rs = np.random.RandomState(5)
    lo, hi = np.linspace(-3, -1, n), np.linspace(1, 4, n)
    don = np.empty((k, P), dtype=np.int32)
    irand = np.empty(P, dtype=np.int32)
    res = np.empty((P, n)) if bounded else None
    r1 = s.random((P, n))
    s.de_async_draws(P, k, n, don, irand, lo, hi, res)
    assert np.array_equal(r1, rs.rand(P, n))
    for i in range(P):
        assert np.array_equal(don[:, i], rs.permutation(np.delete(np.arange(P), i))[:k])
        assert irand[i] == rs.randint(n)
        if bounded:
            assert np.array_equal(res[i], rs.uniform(lo, hi, n))
    assert np.array_equal(s.random((3,)), rs.rand(3))  # both streams stand at the same place
    d = LegacyStream(5).de_generation_async(2, P, n, k, (lo, hi) if bounded else None)
    assert np.array_equal(d["donors"], don) and np.array_equal(d["irand"], irand) and np.array_equal(d["r1"], r1)
    if bounded:
        assert np.array_equal(d["resample"], res)


def test_rng_global_state_interchange(lib):
    """seed=None continues numpy's global stream; sync_back() hands the advanced state back."""
    from stochopy_amd import _rng

    np.random.seed(123)
    a = np.random.rand(3)
    s = _rng.LegacyHostStream(None)
    b = s.random((4,))
    s.sync_back()
    c = np.random.rand(2)
    np.random.seed(123)
    assert np.array_equal(np.concatenate([a, b, c]), np.random.rand(9))


def test_lhs_matches_oracle(lib):
    import oracle
    from oracle import engine as oe

    lo, hi = np.full(5, -2.0), np.linspace(1, 3, 5)
    got = Stream(9).s.latin_hypercube(24, 5, lo, hi)
    want = oe.latin_hypercube(oracle.LegacyStream(9), 24, 5, lo, hi)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("m", [0, 1, 7, 8, 9, 127, 128, 129, 135, 255, 256, 1023, 1024, 6143])
def test_sum_plan_reproduces_numpy_pairwise(lib, m):
    """sx_sum_plan drives the kernels' summation order: replay it in Python and compare with ndarray.sum()."""
    buf = np.zeros(4 + 2 * (m // 64 + 2), dtype=np.int32)
    got = lib.sx_sum_plan(m, buf.ctypes.data, buf.size)
    assert got >= 4
    nleaf, tail, mb, depth = (int(v) for v in buf[:4])
    assert mb == m // 8 and tail == (m % 8 if m >= 8 else m)
    a = np.random.RandomState(m).uniform(-1e3, 1e3, m)
    stack, b0 = [], 0
    cur = 0.0
    for t in range(nleaf):
        b1, merges = int(buf[4 + 2 * t]), int(buf[5 + 2 * t])
        r = [a[b0 * 8 + j] for j in range(8)]
        for b in range(b0 + 1, b1):
            for j in range(8):
                r[j] = r[j] + a[b * 8 + j]
        cur = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
        if t == nleaf - 1:
            for k in range(tail):
                cur = cur + a[b1 * 8 + k]
        stack.append(cur)
        assert len(stack) <= depth
        for _ in range(merges):
            top = stack.pop()
            stack[-1] = stack[-1] + top
        b0 = b1
    if nleaf == 0:
        for k in range(tail):
            cur = cur + a[k]
        stack = [cur]
    assert len(stack) == 1
    assert 0.0 + stack[0] == a.sum()


def test_exchange_buffer_geometry(lib):
    """Host side of the peer exchange: slots[2][8][w] + probe[8][w] words of 8 bytes, w = 2(n+2) rounded up to
    whole 128-byte lines; the relay holds two records + their ready lines; bad arguments are refused."""
    for n in (1, 2, 6, 7, 128, 1024, 2560):
        w = -(-2 * (n + 2) // 16) * 16
        assert lib.sx_xchg_bytes(8, n) == 3 * 8 * w * 8 == lib.sx_xchg_bytes(1, n)
        assert lib.sx_xchg_relay_bytes(n) == 2 * (w + 16) * 8
    assert lib.sx_xchg_bytes(9, 8) == -1 and lib.sx_xchg_bytes(0, 8) == -1 and lib.sx_xchg_bytes(2, 0) == -1
    assert lib.sx_xchg_relay_bytes(0) == -1
    from stochopy_amd import _lib

    x = _lib.SxXchgArgs()
    x.world, x.rank = 2, 5  # rank outside the world
    assert lib.sx_xchg_probe(C.byref(x), 8, 1, None) != 0 and b"world / rank" in lib.sx_last_error()


def test_penalize_host_bookkeeping_follows_the_oracle():
    """The host half of constraints="Penalize" (stochopy_amd.optimize._cmaes._BoundaryWeights) against the
    oracle's restatement (itself pinned to the reference) over a random walk of the mean in and out of the box:
    identical boundary weights / history / flags, and the same penalised fitness."""
    import oracle
    from oracle import engine as oe
    from stochopy_amd.optimize._cmaes import _BoundaryWeights

    rs = np.random.RandomState(5)
    n, P, mueff = 7, 12, 3.4
    ref, got = oe.PenalizeState(n), _BoundaryWeights(n)
    fun = oracle.OBJECTIVES["sphere"]
    xold = np.zeros(n)
    xmean = rs.uniform(-0.5, 0.5, n)
    for it in range(1, 60):
        sigma = 0.05 + 0.3 * rs.rand()
        diagC = rs.uniform(0.2, 3.0, n)
        arx = xmean + sigma * rs.randn(P, n) * np.sqrt(diagC)
        fit_ref, valid = ref.apply(arx, xmean, xold, sigma, diagC, mueff, it, fun)
        raw = fun(np.clip(arx, -1.0, 1.0))
        v = got.update(raw, xmean, xold, sigma, diagC, mueff, it, P)
        assert np.array_equal(got.weights, ref.weights) and np.array_equal(got.spreads, ref.dfithist)
        assert (got.have_spread, got.initial_phase) == (ref.validfitval, ref.iniphase)
        assert np.allclose(raw + ((np.clip(arx, -1.0, 1.0) - arx) ** 2) @ v, fit_ref, rtol=1e-13)
        xold, xmean = xmean, xmean + rs.uniform(-0.4, 0.6, n) * (1.0 if it % 7 else -2.0)
    assert ref.weights.max() > 0.0 and not ref.iniphase  # the walk did leave the box and weights did grow


def test_api_surface_and_validation():
    """Signatures/defaults of the reference (de/_de.py:13-33 etc.) and its bare ValueError/TypeError validation."""
    import inspect

    import stochopy_amd as sa

    sig = inspect.signature(sa.optimize.minimize)
    assert list(sig.parameters) == ["fun", "bounds", "x0", "args", "method", "options", "callback"]
    assert sig.parameters["method"].default == "de"
    de = inspect.signature(sa.optimize.de).parameters
    assert (de["maxiter"].default, de["popsize"].default, de["mutation"].default, de["recombination"].default,
            de["strategy"].default, de["updating"].default, de["xtol"].default) == (100, 10, 0.5, 0.9, "best1bin",
                                                                                    "immediate", 1e-8)
    cp = inspect.signature(sa.optimize.cpso).parameters
    assert (cp["inertia"].default, cp["cognitivity"].default, cp["sociability"].default,
            cp["competitivity"].default) == (0.7298, 1.49618, 1.49618, 1.0)
    assert "competitivity" not in inspect.signature(sa.optimize.pso).parameters
    cm = inspect.signature(sa.optimize.cmaes).parameters
    assert (cm["sigma"].default, cm["muperc"].default) == (0.1, 0.5) and "updating" not in cm
    f, b = sa.factory.rosenbrock, [[-5.12, 5.12]] * 2
    with pytest.raises(TypeError):
        sa.optimize.minimize(42, b)
    # any other Python callable is the reference's fun(x, *args) (SURVEY.md 8b iii): accepted, evaluated per row on
    # the host between the device kernels -- which still needs the GPU (here: the loud no-device error, not a refusal)
    from stochopy_amd._device import NoDeviceError
    from stochopy_amd.optimize import _common

    assert isinstance(_common.resolve_objective(lambda x: float(np.sum(x)), ()), _common.HostExternal)
    # host pools for such an objective: explicit options, or the reference's own spelling (its _common.py:94-97)
    plain = lambda x: float(np.sum(x))  # noqa: E731
    ext = _common.resolve_objective(plain, (), host_workers=3, host_backend="loky")
    assert (ext.pool.workers, ext.pool.backend, ext.from_reference_options) == (3, "loky", False)
    ext = _common.resolve_objective(plain, (), host_workers=2)
    assert ext.pool.backend == "threading"  # the reference's default backend
    ext = _common.resolve_objective(plain, (), workers=5, backend="threading")
    assert (ext.pool.workers, ext.from_reference_options) == (5, True)
    assert _common.resolve_workers(5, ext) == 1 and _common.resolve_backend("threading", ext) == "hip"
    assert _common.resolve_objective(plain, (), workers=1, backend="loky").pool is None  # serial, as in the reference
    assert _common.resolve_objective(plain, (), host_workers=-1).pool.workers == (os.cpu_count() or 1) or (os.cpu_count() or 1) < 2
    with pytest.raises(ValueError):
        _common.resolve_objective(plain, (), host_workers=4, host_backend="mpi")
    with pytest.raises(ValueError):
        _common.resolve_objective(plain, (), workers=4, backend="loky", host_workers=2)
    with pytest.raises(ValueError):  # a factory objective runs in the kernels: no host pool to configure
        _common.resolve_objective(f, (), host_workers=4)
    with pytest.raises(ValueError, match="belongs to keurfonluu/stochopy"):
        sa.optimize.minimize(f, b, options={"backend": "loky", "workers": 4})
    # the pool's tasks are the reference's serial wrapper on blocks of rows
    rows = np.arange(12.0).reshape(4, 3)
    for backend in ("threading", "loky"):
        pool = _common.HostPool(2, backend)
        got = pool.executor().submit(_common._eval_block, np.sum, rows, ()).result()
        assert np.array_equal(got, rows.sum(axis=1))
        pool.close()
    with pytest.raises(NoDeviceError):
        sa.optimize.minimize(lambda x: float(np.sum(x)), b, options={"backend": "hip", "updating": "deferred"})
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, [-1, 1])
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, b, options={"popsize": 1})
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, b, options={"mutation": 3.0})
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, b, method="cpso", options={"inertia": 1.5})
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, b, method="cmaes", options={"sigma": 0.0})
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, b, options={"backend": "cuda"})
    with pytest.raises(ValueError):
        sa.optimize.minimize(f, b, options={"backend": "loky"})
    with pytest.raises(KeyError):
        sa.optimize.minimize(f, b, method="sampler")  # not an optimizer of the reference's registry


def test_optimize_result_surface():
    """stochopy/_common.py:1-35: dict with attribute access; repr sorted, right-justified, hides xall/funall."""
    from stochopy_amd.optimize import OptimizeResult

    r = OptimizeResult(x=np.array([1.0]), fun=2.0, nit=3, xall=np.zeros(3), funall=np.zeros(3))
    assert r.fun == 2.0 and r["nit"] == 3 and sorted(dir(r)) == ["fun", "funall", "nit", "x", "xall"]
    # width = longest key (incl. the hidden ones) + 1, as in the reference
    assert repr(r).splitlines()[0] == "    fun: 2.0" and "xall" not in repr(r)
    with pytest.raises(AttributeError):
        r.missing
    assert repr(OptimizeResult()) == "OptimizeResult()"


def test_no_gpu_means_loud_failure():
    """Without a ROCm device the product path must raise -- there is no CPU fallback."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    import stochopy_amd as sa
    from stochopy_amd._device import NoDeviceError

    with pytest.raises(NoDeviceError):
        sa.optimize.minimize(sa.factory.sphere, [[-1, 1]] * 3, options={"maxiter": 3, "popsize": 8, "seed": 0})
    with pytest.raises(NoDeviceError):
        sa.factory.sphere(np.ones(4))


def test_product_never_imports_the_oracle():
    for root, _, files in os.walk(os.path.join(ROOT, "stochopy_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp")):
                text = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f


def test_workers_option_resolution():
    """`workers` = number of GPUs; None / 0 / 1 = one; -1 = all ranks of the process group (none here -> 1)."""
    from stochopy_amd.optimize import _common

    assert [_common.resolve_workers(w) for w in (None, 0, 1, -1, 4)] == [1, 1, 1, 1, 4]
    with pytest.raises(ValueError):
        _common.resolve_workers(-3)


def test_chained_generations_are_planned_as_few_graph_replays():
    """_DeRun.plan_chain (host logic of the chained DE engine): whole 50-generation graphs, a repeated remainder as one
    graph of exactly that even length, a one-off remainder from the 10-generation graph, single launches for the rest;
    the generations always add up and graphs start at the parity they were built for."""
    from stochopy_amd.optimize._de import _DeRun

    plan = _DeRun.plan_chain
    assert plan(200, 0, {}, 50, 10) == [50] * 4
    assert plan(137, 0, {}, 50, 10) == [50, 50, 10, 10, 10] + [0] * 7  # first request: no new graph for 36
    assert plan(9, 3, {}, 50, 10) == [0] * 9
    seen = {}
    assert plan(20, 5, seen, 50, 10) == [10, 10]      # the driver's --steps 20 --warmup 5: first block ...
    assert plan(20, 25, seen, 50, 10) == [20]         # ... and every later one (odd parity, same length)
    assert plan(20, 45, seen, 50, 10) == [20]
    assert plan(20, 64, seen, 50, 10) == [10, 10]     # the other parity is a different graph: counted on its own
    assert plan(20, 84, seen, 50, 10) == [20]
    assert plan(71, 0, seen, 50, 10) == [50, 20, 0]   # 21 left after the whole graph: the (parity 0, 20) graph exists
    rs = np.random.RandomState(3)
    seen, launches = {}, 0
    for _ in range(300):
        n = int(rs.randint(0, 260))
        p = plan(n, launches, seen, 50, 10)
        assert sum(s if s else 1 for s in p) == n
        at = launches
        for s in p:
            assert s == 0 or (s % 2 == 0 and 10 <= s <= 50)
            at += s if s else 1
        launches += n
    assert all(k[1] % 2 == 0 and 10 < k[1] < 50 for k in seen)


@pytest.mark.parametrize("P,n", [(2, 3), (7, 5), (100, 70), (513, 2), (4096, 128)])
def test_philox_latin_hypercube_oracle_properties(P, n):
    """The counter-based initial population of the throughput mode (oracle side; the HIP kernel is compared with it bit
    for bit in tests/test_gpu_de.py): it IS a Latin hypercube in the reference's sense (_common.py:109-120) -- every
    column visits every stratum of width (upper - lower) / P exactly once, the jitter fills the LOWER HALF of the cell
    (rand / P inside strata 2 / P wide) -- and any block of rows can be drawn on its own (sharded runs)."""
    import oracle
    from oracle import engine as oe

    s = oracle.PhiloxStream(99)
    lo, up = np.full(n, -5.12), np.linspace(1.0, 5.12, n)
    X = s.lhs_population(P, n, lo, up)
    cell = (X - lo) / (up - lo) * P
    strata = np.floor(cell).astype(int)
    assert all(sorted(strata[:, j]) == list(range(P)) for j in range(n))
    assert (cell - strata < 0.5 + 1e-9).all()
    parts = [s.lhs_population(P, n, lo, up, row0=r0, rows=min(3, P - r0)) for r0 in range(0, P, 3)]
    assert np.array_equal(np.vstack(parts), X)
    assert not np.array_equal(X, oracle.PhiloxStream(100).lhs_population(P, n, lo, up))
    assert np.array_equal(oe.latin_hypercube(oracle.PhiloxStream(99), P, n, lo, up), X)  # what oracle.minimize starts from


def test_replicated_workers_says_so_once_per_method():
    """workers > 1 where a run cannot be sharded (optimize/_common.py replicated_workers): one warning per method, workers -> 1."""
    import warnings

    from stochopy_amd.optimize import _common

    _common._warned_replicated.discard("unit-test")
    with pytest.warns(UserWarning, match="replicated"):
        assert _common.replicated_workers("unit-test", 4, "a reason") == 1
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert _common.replicated_workers("unit-test", 4, "a reason") == 1
