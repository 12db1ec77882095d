"""The N>1 path: population sharded by rows, one process per rank, one record all-gather per generation."""
import os
import socket
import tempfile

import numpy as np
import pytest

import oracle
from oracle import engine as oe


_used_ports = set()


def _free_port():
    """A rendezvous port nobody listens on and that this session has not handed out before (the kernel likes to
    give the same ephemeral port again right after a process group went away, while its store may still be closing)."""
    for _ in range(64):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        if port not in _used_ports:
            _used_ports.add(port)
            return port
    raise RuntimeError("no free rendezvous port")


def _spawn(fn, world, cfg):
    import torch.multiprocessing as mp

    for attempt in range(3):
        out = tempfile.mkdtemp(prefix="sx_dist_")
        try:
            mp.spawn(fn, args=(world, _free_port(), cfg, out), nprocs=world, join=True)
            return out
        except Exception as e:  # noqa: BLE001  rendezvous trouble (port taken meanwhile, store socket reset): again
            # Only genuine rendezvous errors are retried.  A worker that dies on a signal (SIGSEGV / SIGABRT / SIGBUS)
            # FAILS the test: that is what a memory fault in the peer exchange, a graph teardown or an RCCL capture
            # would look like.
            rendezvous = any(w in str(e) for w in ("EADDRINUSE", "DistNetworkError", "TCPStore", "Connection reset",
                                                   "Broken pipe", "failed to listen", "failed to connect"))
            if "terminated with signal" in str(e) or not rendezvous or attempt == 2:
                raise


def _sharded_oracle(cfg, world):
    n = cfg["n"]
    return oe.run_de_sharded(oracle.OBJECTIVES["rosenbrock"], np.full(n, -5.12), np.full(n, 5.12),
                             oracle.PhiloxStream(cfg["seed"]), world, maxiter=cfg["gens"], popsize=cfg["P"],
                             ftol=-1.0, xtol=0.0)


def test_shard_bounds_and_record_rule():
    from stochopy_amd import parallel

    assert [parallel.shard_bounds(4096, 8, r) for r in (0, 7)] == [(0, 512), (3584, 512)]
    # any popsize (the reference's MPI loop takes any, _common.py:64-65): blocks of ceil(P / world) rows, the last rank short
    assert [parallel.shard_bounds(10, 4, r) for r in range(4)] == [(0, 3), (3, 3), (6, 3), (9, 1)]
    assert [parallel.shard_bounds(131073, 8, r) for r in (0, 7)] == [(0, 16385), (114695, 16378)]
    with pytest.raises(ValueError):
        parallel.shard_bounds(5, 4, 0)  # blocks of 2 leave the last rank nothing
    rec = np.array([[2.0, 700.0, 0, 0], [1.0, 900.0, 1, 1], [1.0, 300.0, 2, 2], [3.0, 0.0, 3, 3]])
    assert parallel.best_of_records(rec) == (2, 1.0, 300)  # ties -> smallest global row = np.argmin's first minimum


def test_world_exchange_two_ranks_gloo_cpu():
    """world_size 2 on CPU: the product's World exchange vs the one-process simulation of the same semantics."""
    from _dist_workers import cpu_exchange_worker

    for cfg in ({"n": 12, "P": 48, "gens": 8, "seed": 31337}, {"n": 12, "P": 49, "gens": 8, "seed": 31338}):  # 24 + 24, 25 + 24 rows
        out = _spawn(cpu_exchange_worker, 2, cfg)
        ref = _sharded_oracle(cfg, 2)
        for r in range(2):
            assert np.array_equal(np.load(os.path.join(out, f"trace_{r}.npy")), np.array(ref["_trace"]))


def test_world_row_gather_and_agreement_gloo_cpu():
    """The set-up / CMA-ES plumbing of parallel.World with real processes (world_size 2, CPU)."""
    from _dist_workers import cpu_rows_worker

    out = _spawn(cpu_rows_worker, 2, {"rows": 3, "n": 4})
    want = np.concatenate([np.arange(12.0).reshape(3, 4) + 1000.0 * r for r in range(2)])
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(out, f"rows_{r}.npy")), want)
        assert np.array_equal(np.load(os.path.join(out, f"fit_{r}.npy")), want[:, 0])
    # a population the ranks do not divide: 7 rows over 3 ranks = blocks of 3, 3, 1 (padded on the way, parallel.World)
    out = _spawn(cpu_rows_worker, 3, {"total": 7, "n": 4})
    want = np.concatenate([(np.arange(12.0).reshape(3, 4) + 1000.0 * r)[:c] for r, c in enumerate((3, 3, 1))])
    for r in range(3):
        assert np.array_equal(np.load(os.path.join(out, f"rows_{r}.npy")), want)
        assert np.array_equal(np.load(os.path.join(out, f"fit_{r}.npy")), want[:, 0])


@pytest.mark.parametrize("constraints", [None, "Shrink"])
def test_cpso_fit_radius_gather_two_ranks_gloo_cpu(constraints):
    """world_size 2 on CPU: the competitive restart's [pbestfit | radii] gather (optimize/_cpso.py `fit_radius_all`) and
    the best-record exchange through parallel.World -- the sharded swarm IS the unsharded oracle run: same best-f trace,
    same restarts (generation, nw) and the same re-seeded rows on every rank."""
    from _dist_workers import cpu_cpso_worker

    cfg = {"n": 4, "P": 40, "maxiter": 300, "gamma": 1.0, "npart": 3, "seed": 77, "objective": "ackley",
           "constraints": constraints, "xtol": 1e-12, "ftol": 1e-12}
    out = _spawn(cpu_cpso_worker, 2, cfg)
    n = cfg["n"]
    ref = oe.run_pso(oracle.OBJECTIVES["ackley"], np.full(n, -32.768), np.full(n, 32.768), None, oracle.PhiloxStream(77),
                     maxiter=300, popsize=40, competitivity=1.0, xtol=1e-12, ftol=1e-12, constraints=constraints)
    assert len(ref["_restarts"]) >= 2  # the case must exercise the restart
    for r in range(2):
        got = np.load(os.path.join(out, f"cpso_{r}.npz"))
        assert int(got["nit"]) == ref["nit"] and int(got["status"]) == ref["status"]
        assert float(got["fun"]) == ref["fun"] and np.array_equal(got["x"], ref["x"])
        assert [tuple(v) for v in got["restarts"].astype(int)] == ref["_restarts"]
        assert np.array_equal(got["rows"], np.concatenate(ref["_restart_rows"]))


@pytest.mark.parametrize("method,constraints", [("cmaes", None), ("cmaes", "Penalize"), ("vdcma", None)])
def test_cma_candidate_gather_two_ranks_gloo_cpu(method, constraints):
    """world_size 2 on CPU: CMA-ES / VD-CMA shard the candidates (rows drawn and evaluated by their owner, gathered with
    World.all_gather_rows, model update replicated) -- the unsharded oracle run bit for bit on every rank."""
    from _dist_workers import cpu_cma_worker

    P = 12 if constraints is None else 13  # (13 over 2 ranks: 7 + 6 rows, the last rank short)
    cfg = {"n": 5, "P": P, "maxiter": 30, "seed": 9, "objective": "rosenbrock", "constraints": constraints, "method": method}
    out = _spawn(cpu_cma_worker, 2, cfg)
    run = oe.run_vdcma if method == "vdcma" else oe.run_cmaes
    ref = run(oracle.OBJECTIVES["rosenbrock"], np.full(5, -3.0), np.full(5, 3.0), None, oracle.PhiloxStream(9), maxiter=30,
              popsize=P, sigma=0.3, constraints=constraints, eigh="canonical")
    for r in range(2):
        got = np.load(os.path.join(out, f"cma_{r}.npz"))
        assert int(got["nit"]) == ref["nit"] and int(got["status"]) == ref["status"] and int(got["calls"]) >= ref["nit"]
        assert float(got["fun"]) == ref["fun"] and np.array_equal(got["x"], ref["x"])


def _de_reference(cfg, world):
    o = dict(cfg["options"])
    n = cfg["n"]
    seed = o.pop("seed")
    o.pop("exchange", None)
    o.pop("donors", None)
    return oe.run_de_sharded(oracle.OBJECTIVES[cfg["objective"]], np.full(n, -5.12), np.full(n, 5.12),
                             oracle.PhiloxStream(seed), world, **o)


def _check_sharded_de(world, cfg, worker=None):
    from _dist_workers import gpu_minimize_worker

    out = _spawn(worker or gpu_minimize_worker, world, cfg)
    ref = _de_reference(cfg, world)
    for r in range(world):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status), (r, fun, nit, status, ref)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)
        assert open(os.path.join(out, f"exchange_{r}.txt")).read() == cfg["options"]["exchange"]


def _de_cfg(n, P, gens, seed, exchange, objective="rosenbrock", **more):
    opts = {"maxiter": gens, "popsize": P, "seed": seed, "ftol": -1.0, "xtol": 0.0, "exchange": exchange}
    opts.update(more)
    return {"n": n, "objective": objective, "method": "de", "options": opts}


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_de_on_gpu_matches_oracle(world, exchange):
    """Ranks share the single test GPU (gloo for set-up); sharded HIP DE == oracle.run_de_sharded bit for bit,
    with the record exchange staged through the process group ("rccl" path) and with the generation kernels
    writing into each other's IPC-mapped exchange buffers ("p2p")."""
    _check_sharded_de(world, _de_cfg(24, 128, 9, 2024, exchange))


@pytest.mark.gpu
def test_c5_full_size_sharded_over_eight_ranks_matches_oracle():
    """BASELINE config 5 in its sharded form AT SIZE: DE n=1024, P=131072 over 8 ranks (16384 rows each; here the
    ranks share the one test GPU, so the records travel between kernels: exchange="rccl"), shard-local donors, the
    initial evaluation + 2 generations == oracle.run_de_sharded bit for bit on every rank (x, f, nit, status)."""
    _check_sharded_de(8, _de_cfg(1024, 131072, 3, 2026, "rccl"))


@pytest.mark.gpu
@pytest.mark.parametrize("case", [
    dict(n=128, P=64, gens=130, seed=5),                                   # bounds-test-free kernel, graph replays
    dict(n=300, P=36, gens=60, seed=6),                                    # whole wave per row, several leaves
    dict(n=10, P=40, gens=30, seed=7, strategy="rand1bin"),                # the row of the record is never read
    dict(n=33, P=48, gens=25, seed=8, strategy="best2bin", constraints="Random"),
    dict(n=16, P=64, gens=400, seed=9, objective="sphere", ftol=1e-3, xtol=1e-8),   # stops on ftol (status 0/1)
    dict(n=2048, P=2200, gens=5, seed=10),                                 # 550 workgroup records per shard (> 512)
    dict(n=3000, P=40, gens=5, seed=11),        # rows of 2049 ... 4096 elements: the chained kernel for the run's duration
    dict(n=24, P=131, gens=9, seed=12),         # a population the ranks do not divide: 66 + 65 rows
])
def test_p2p_exchange_cases(case):
    case = dict(case)
    cfg = _de_cfg(case.pop("n"), case.pop("P"), case.pop("gens"), case.pop("seed"), "p2p", **case)
    _check_sharded_de(2, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("world,case", [
    (2, dict(n=24, P=128, gens=30, seed=77)),
    (4, dict(n=24, P=128, gens=30, seed=77)),
    (2, dict(n=13, P=40, gens=20, seed=3, strategy="rand2bin", constraints="Random")),   # 5 donors, two Philox calls
    (2, dict(n=300, P=48, gens=12, seed=4, strategy="rand1bin")),                         # whole-wave rows
    (4, dict(n=128, P=256, gens=60, seed=5)),                                             # bounds-test-free kernel
    # BASELINE config 5's row length and rank count; the population is cut to what 8 ranks can keep co-resident on the
    # ONE test GPU (ranks that share a device must all fit at once: a rank's waiting workgroups may otherwise fill it
    # before a peer's record-pushing workgroup is resident -- with one GPU per rank that cannot happen)
    (8, dict(n=1024, P=2048, gens=4, seed=6)),
    (2, dict(n=3000, P=24, gens=6, seed=12)),    # rows of 2049 ... 4096 elements (refused in round 5)
    (4, dict(n=24, P=130, gens=12, seed=13)),    # 33 + 33 + 33 + 31 rows: the owner of a donor row is row // 33
])
def test_global_donors_reproduce_the_unsharded_run(world, case):
    """donors="global": donor rows are drawn over the whole population and read from their owners' HBM
    (IPC-mapped population buffers) inside the generation kernel -- the sharded run IS the unsharded run."""
    from _dist_workers import gpu_minimize_worker

    case = dict(case)
    n, P, gens, seed = case.pop("n"), case.pop("P"), case.pop("gens"), case.pop("seed")
    cfg = _de_cfg(n, P, gens, seed, "p2p", donors="global", **case)
    out = _spawn(gpu_minimize_worker, world, cfg)
    opts = {k: v for k, v in cfg["options"].items() if k not in ("exchange", "donors")}
    ref = oracle.minimize("rosenbrock", [[-5.12, 5.12]] * n, method="de", options=dict(opts, updating="deferred"),
                          rng="philox")
    for r in range(world):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)


@pytest.mark.gpu
@pytest.mark.parametrize("method,extra", [("de", {"exchange": "p2p"}), ("de", {"exchange": "p2p", "donors": "global"}),
                                          ("de", {"exchange": "rccl"}), ("pso", {})])
def test_eight_ranks_share_one_gpu(method, extra):
    """The full node's world size (8 ranks: one pushing wavefront per peer, 32 header words per wait) on the one
    test GPU, DE through both transports and with global donors, PSO through the exchange kernel."""
    from _dist_workers import gpu_minimize_worker

    n, P, gens, seed = 128, 256, 25, 41
    cfg = _de_cfg(n, P, gens, seed, extra.get("exchange", "p2p"), **{k: v for k, v in extra.items() if k != "exchange"})
    cfg["method"] = method
    if method == "pso":
        cfg["options"].pop("exchange")
        cfg["env"] = {"SX_EXCHANGE": "p2p"}
    out = _spawn(gpu_minimize_worker, 8, cfg)
    opts = {k: v for k, v in cfg["options"].items() if k not in ("exchange", "donors")}
    if method == "de" and extra.get("donors") != "global":
        ref = _de_reference(cfg, 8)
    else:
        ref = oracle.minimize("rosenbrock", [[-5.12, 5.12]] * n, method=method, options=dict(opts, updating="deferred"),
                              rng="philox")
    for r in range(8):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)


@pytest.mark.gpu
def test_p2p_exchange_timeout_is_reported():
    """A rank that never shows up: the others give up after the timeout and raise (no hang)."""
    from _dist_workers import gpu_p2p_straggler_worker

    out = _spawn(gpu_p2p_straggler_worker, 2, _de_cfg(24, 128, 9, 2024, "p2p"))
    assert "timed out" in open(os.path.join(out, "err_0.txt")).read()


@pytest.mark.gpu
@pytest.mark.timeout(300)
@pytest.mark.parametrize("method", ["de", "pso"])
def test_late_failure_on_one_rank_raises_on_every_rank(method):
    """One rank's callback raises at the last generation (its peers have already finished their loop): all ranks leave
    minimize() with an exception and meet again afterwards -- the success path and the failure path issue the SAME
    collective (one all_gather of the success flags), no barrier that a failing rank would skip."""
    from _dist_workers import gpu_late_failure_worker

    opts = {"maxiter": 12, "popsize": 64, "seed": 3, "ftol": -1.0, "xtol": 0.0}
    if method == "de":
        opts["exchange"] = "p2p"  # (PSO takes the transport from the environment)
    cfg = {"n": 12, "objective": "rosenbrock", "method": method, "options": opts, "env": {"SX_EXCHANGE": "p2p"}}
    out = _spawn(gpu_late_failure_worker, 2, cfg)
    assert "callback failed on purpose" in open(os.path.join(out, "err_1.txt")).read()
    assert "peer rank failed" in open(os.path.join(out, "err_0.txt")).read()


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
def test_sharded_pso_on_gpu_is_exact(exchange):
    """PSO state is row-local and the Philox counters use global rows: 2 shards == the unsharded run, with the
    best record travelling through the process group or by peer writes (one-workgroup exchange kernel)."""
    from _dist_workers import gpu_minimize_worker

    opts = {"maxiter": 40, "popsize": 96, "seed": 77, "ftol": -1.0, "xtol": 0.0}
    cfg = {"n": 20, "objective": "rosenbrock", "method": "pso", "options": opts, "env": {"SX_EXCHANGE": exchange}}
    out = _spawn(gpu_minimize_worker, 2, cfg)
    ref = oracle.minimize("rosenbrock", [[-5.12, 5.12]] * 20, method="pso", options=dict(opts, updating="deferred"), rng="philox")
    for r in range(2):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)
        assert open(os.path.join(out, f"exchange_{r}.txt")).read() == exchange


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_cpso_restart_is_exact(world, exchange):
    """Competitive restart over a sharded swarm: the radius (a max) and the worst-nw rule (a rank) span ALL
    particles -- one all-gather of [pbestfit | partial radii] per generation -- and the new positions are keyed
    by the global row, so the shards reproduce the unsharded run.  Restarts must actually fire."""
    from _dist_workers import gpu_minimize_worker

    opts = {"maxiter": 30, "popsize": 256, "seed": 5, "ftol": -1.0, "xtol": 0.0}
    cfg = {"n": 16, "objective": "sphere", "method": "cpso", "options": opts, "env": {"SX_EXCHANGE": exchange}}
    out = _spawn(gpu_minimize_worker, world, cfg)
    ref = oracle.minimize("sphere", [[-5.12, 5.12]] * 16, method="cpso", options=dict(opts, updating="deferred"), rng="philox")
    assert len(ref["_restarts"]) > 0
    for r in range(world):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)


@pytest.mark.gpu
@pytest.mark.parametrize("method,extra,env", [
    ("pso", {}, {"SX_EXCHANGE": "rccl"}),
    ("cpso", {"constraints": "Shrink", "inertia": 0.91}, {"SX_EXCHANGE": "p2p"}),
    ("de", {"exchange": "p2p", "donors": "global", "strategy": "rand1bin"}, {}),
    ("de", {"exchange": "p2p", "donors": "global", "verbosity": 0.0}, {}),
])
@pytest.mark.parametrize("with_callback", [False, True])
def test_sharded_callbacks_and_return_all(method, extra, env, with_callback):
    """workers > 1 with return_all / callback: every rank sees the WHOLE population each generation (as the
    reference's parallel backends do).  PSO / CPSO shards and DE with global donors are the unsharded run, so
    xall, funall and everything the callback receives must equal the oracle's, on every rank."""
    from _dist_workers import gpu_minimize_worker

    n, world = 12, 2
    opts = dict({"maxiter": 25, "popsize": 64, "seed": 21, "ftol": -1.0, "xtol": 0.0, "return_all": True}, **extra)
    cfg = {"n": n, "objective": "sphere" if method == "cpso" else "rosenbrock", "method": method, "options": opts,
           "env": env, "callback": with_callback}
    out = _spawn(gpu_minimize_worker, world, cfg)
    seen = []
    oopts = {k: v for k, v in opts.items() if k not in ("exchange", "donors")}
    ref = oracle.minimize(cfg["objective"], [[-5.12, 5.12]] * n, method=method, options=dict(oopts, updating="deferred"), rng="philox",
                          callback=lambda X, r: seen.append((X.copy(), float(r.fun), int(r.nit), int(r.nfev))))
    for r in range(world):
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)
        assert np.array_equal(np.load(os.path.join(out, f"xall_{r}.npy")), ref.xall)
        assert np.array_equal(np.load(os.path.join(out, f"funall_{r}.npy")), ref.funall)
        if with_callback:
            assert np.array_equal(np.load(os.path.join(out, f"cbX_{r}.npy")), np.array([c[0] for c in seen]))
            assert np.array_equal(np.load(os.path.join(out, f"cbmeta_{r}.npy")), np.array([c[1:] for c in seen]))


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["pso", "cpso", "de"])
def test_sharded_run_with_a_caller_supplied_objective(method):
    """workers=2 around a caller-supplied device objective that returns the fused kernel's values (sx_eval handed
    in through factory.batched): sharded PSO / CPSO must again be the unsharded oracle run, and sharded DE the
    sharded oracle -- and around a torch-written objective both ranks must agree."""
    from _dist_workers import gpu_minimize_worker

    n = 10
    opts = {"maxiter": 20, "popsize": 64, "seed": 8, "ftol": -1.0, "xtol": 0.0}
    cfg = {"n": n, "objective": "sphere", "method": method, "options": opts, "external": "device-sphere",
           "env": {"SX_EXCHANGE": "rccl"}}
    out = _spawn(gpu_minimize_worker, 2, cfg)
    if method == "de":
        ref = oe.run_de_sharded(oracle.OBJECTIVES["sphere"], np.full(n, -5.12), np.full(n, 5.12), oracle.PhiloxStream(8),
                                2, maxiter=20, popsize=64, ftol=-1.0, xtol=0.0)
    else:
        ref = oracle.minimize("sphere", [[-5.12, 5.12]] * n, method=method, options=dict(opts, updating="deferred"), rng="philox")
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)
        assert np.load(os.path.join(out, f"meta_{r}.npy"))[0] == ref.fun
    out = _spawn(gpu_minimize_worker, 2, dict(cfg, external="batched"))
    assert np.array_equal(np.load(os.path.join(out, "x_0.npy")), np.load(os.path.join(out, "x_1.npy")))
    assert np.load(os.path.join(out, "meta_0.npy"))[0] < 10.0


@pytest.mark.gpu
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
def test_sharded_de_island_model_with_return_all(exchange):
    """Shard-local donors (the default island model) with return_all + callback: same best as the run without
    them, and the recorded rows are the gathered shards (first rows = rank 0's)."""
    from _dist_workers import gpu_minimize_worker

    cfg = _de_cfg(24, 128, 9, 2024, exchange)
    plain = _spawn(gpu_minimize_worker, 2, cfg)
    cfg2 = _de_cfg(24, 128, 9, 2024, exchange, return_all=True)
    cfg2["callback"] = True
    out = _spawn(gpu_minimize_worker, 2, cfg2)
    for r in range(2):
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), np.load(os.path.join(plain, f"x_{r}.npy")))
        xall, funall = np.load(os.path.join(out, f"xall_{r}.npy")), np.load(os.path.join(out, f"funall_{r}.npy"))
        assert xall.shape == (9, 128, 24) and funall.shape == (9, 128)
        cbX = np.load(os.path.join(out, f"cbX_{r}.npy"))
        assert cbX.shape == (9, 128, 24) and np.array_equal(cbX, xall)  # full verbosity: the callback's X is xall[it-1]
    assert np.array_equal(np.load(os.path.join(out, "xall_0.npy")), np.load(os.path.join(out, "xall_1.npy")))


@pytest.mark.gpu
@pytest.mark.parametrize("rng", ["philox", "numpy-legacy"])
def test_sharded_cmaes_matches_single_gpu(rng):
    """CMA-ES with workers=2: candidates sampled / evaluated by shards, model update replicated == workers=1."""
    import stochopy_amd as sa
    from _dist_workers import gpu_minimize_worker

    n = 20
    opts = {"maxiter": 40, "popsize": 48, "seed": 11, "rng": rng}
    cfg = {"n": n, "objective": "rosenbrock", "method": "cmaes", "options": opts, "rng": rng}
    # the sharded run is driven by the host loop; with a callback so is the single-GPU run (bit-identical), without
    # one the Philox run stays on the device (csrc/sx_cma_loop.hip: other summation orders, same run to rounding)
    one = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="cmaes",
                               options=dict(opts, backend="hip"), callback=lambda X, r: None)
    resident = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="cmaes",
                                    options=dict(opts, backend="hip"))
    assert (resident.nit, resident.status) == (one.nit, one.status) and np.isclose(resident.fun, one.fun, rtol=1e-6)
    # round 3: the sharded Philox run stays on the device too (sx_cmaes_generation_stage: own candidates, one gather, the
    # model update replicated) -- the SAME kernels as the one-GPU resident run, so bit-identical to it; legacy draws take
    # the host-driven loop on any number of ranks
    want = resident if rng == "philox" else one
    out = _spawn(gpu_minimize_worker, 2, cfg)
    for r in range(2):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (want.fun, want.nit, want.nfev, want.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), want.x)


@pytest.mark.gpu
@pytest.mark.parametrize("rng", ["philox", "numpy-legacy"])
def test_sharded_vdcma_matches_single_gpu(rng):
    """VD-CMA with workers=2 (candidates sampled / evaluated by shards, incl. the injected pair in rank 0's
    shard) == workers=1."""
    import stochopy_amd as sa
    from _dist_workers import gpu_minimize_worker

    n = 30
    opts = {"maxiter": 40, "popsize": 20, "seed": 8, "sigma": 0.25, "rng": rng}
    cfg = {"n": n, "objective": "rosenbrock", "method": "vdcma", "options": opts}
    # the sharded run is driven by the host loop; with a callback so is the single-GPU run (bit-identical), without
    # one the Philox run stays on the device (csrc/sx_vd_loop.hip: other summation orders, same run to rounding)
    one = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma",
                               options=dict(opts, backend="hip"), callback=lambda X, r: None)
    resident = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma",
                                    options=dict(opts, backend="hip"))
    assert (resident.nit, resident.status) == (one.nit, one.status) and np.isclose(resident.fun, one.fun, rtol=1e-6)
    want = resident if rng == "philox" else one  # (round 3: the sharded Philox run stays on the device, as CMA-ES's)
    out = _spawn(gpu_minimize_worker, 2, cfg)
    for r in range(2):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (want.fun, want.nit, want.nfev, want.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), want.x)


@pytest.mark.gpu
def test_sharded_cmaes_penalize_matches_single_gpu():
    """The same with constraints="Penalize" and a box whose optimum lies outside: raw fitness, weights and the
    penalty term are settled per generation across the shards (two extra all-gathers while weights are active)."""
    import stochopy_amd as sa
    from _dist_workers import gpu_minimize_worker

    n = 6
    bounds = [[1.0, 5.0]] * n
    opts = {"maxiter": 60, "popsize": 12, "seed": 99, "sigma": 0.3, "constraints": "Penalize", "rng": "philox"}
    cfg = {"n": n, "objective": "sphere", "method": "cmaes", "options": opts, "bounds": bounds}
    one = sa.optimize.minimize(sa.factory.sphere, bounds, method="cmaes", options=dict(opts, backend="hip"))
    out = _spawn(gpu_minimize_worker, 2, cfg)
    for r in range(2):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (one.fun, one.nit, one.nfev, one.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), one.x)


@pytest.mark.gpu
def test_sharded_path_over_rccl_single_rank():
    """backend "nccl" (RCCL) with one rank: the device-side all_gather_into_tensor exchange used in production."""
    from _dist_workers import nccl_single_rank_worker

    _check_sharded_de(1, _de_cfg(24, 128, 9, 2024, "rccl"), worker=nccl_single_rank_worker)


@pytest.mark.gpu
def test_rccl_path_graph_capture_single_rank():
    """130 generations: two replays of the captured 50-generation graph (kernels + RCCL all-gathers) + 29 eager."""
    from _dist_workers import nccl_single_rank_worker

    _check_sharded_de(1, _de_cfg(24, 128, 130, 77, "rccl"), worker=nccl_single_rank_worker)


@pytest.mark.gpu
def test_rccl_path_blocks_of_twenty_generations_replay_a_captured_graph():
    """`bench.py --gpus N --steps 20` steps the RCCL transport in blocks of 20 generations (< GRAPH_CHUNK): from the second
    block on a block is ONE replay of a captured 20-generation graph (round 5; it used to be 20 eager generations of two
    library calls and a collective each).  Same population and best as 120 generations enqueued at once."""
    from _dist_workers import nccl_single_rank_blocks_worker

    out = _spawn(nccl_single_rank_blocks_worker, 1, {"n": 24, "P": 128, "K": 20, "blocks": 6, "seed": 5})
    it, _ = np.load(os.path.join(out, "blocks_0.npy"))
    assert int(it) == 121


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["pso", "cpso"])
def test_sharded_pso_rccl_graph_capture_single_rank(method):
    """PSO / CPSO over RCCL with one rank, 70 generations: replays of the captured 16-generation graph
    (generation, shard best, all-gather, finalise [, radius, all-gather, select, apply]) == the oracle."""
    from _dist_workers import nccl_single_rank_worker

    opts = {"maxiter": 70, "popsize": 256, "seed": 5, "ftol": -1.0, "xtol": 0.0}
    cfg = {"n": 16, "objective": "sphere", "method": method, "options": opts, "env": {"SX_EXCHANGE": "rccl"},
           "flight_recorder": method == "pso"}  # pso: the capture waits on torch's flight recorder; cpso: it sleeps
    out = _spawn(nccl_single_rank_worker, 1, cfg)
    ref = oracle.minimize("sphere", [[-5.12, 5.12]] * 16, method=method, options=dict(opts, updating="deferred"), rng="philox")
    fun, nit, nfev, status = np.load(os.path.join(out, "meta_0.npy"))
    assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
    assert np.array_equal(np.load(os.path.join(out, "x_0.npy")), ref.x)


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["de", "pso", "cpso"])
def test_callbacks_and_return_all_over_rccl_single_rank(method):
    """The shard gathers behind callback / return_all through the production collective (all_gather_into_tensor on
    the device, backend nccl = RCCL) with one rank: history and callback populations == the oracle's."""
    from _dist_workers import nccl_single_rank_worker

    n = 10
    opts = {"maxiter": 20, "popsize": 48, "seed": 4, "ftol": -1.0, "xtol": 0.0, "return_all": True}
    if method == "de":
        opts["exchange"] = "rccl"
    cfg = {"n": n, "objective": "rosenbrock", "method": method, "options": opts, "callback": True,
           "env": {"SX_EXCHANGE": "rccl"}}
    out = _spawn(nccl_single_rank_worker, 1, cfg)
    seen = []
    oopts = {k: v for k, v in opts.items() if k != "exchange"}
    ref = oracle.minimize("rosenbrock", [[-5.12, 5.12]] * n, method=method, options=dict(oopts, updating="deferred"), rng="philox",
                          callback=lambda X, r: seen.append(X.copy()))
    assert np.array_equal(np.load(os.path.join(out, "x_0.npy")), ref.x)
    assert np.array_equal(np.load(os.path.join(out, "xall_0.npy")), ref.xall)
    assert np.array_equal(np.load(os.path.join(out, "funall_0.npy")), ref.funall)
    assert np.array_equal(np.load(os.path.join(out, "cbX_0.npy")), np.array(seen))


@pytest.mark.gpu
def test_p2p_path_single_rank_nccl_setup():
    """Same with the peer exchange (handles and agreement travel over the RCCL group, the kernels write locally)."""
    from _dist_workers import nccl_single_rank_worker

    _check_sharded_de(1, _de_cfg(24, 128, 9, 2024, "p2p"), worker=nccl_single_rank_worker)


def _n_gpus():
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:  # noqa: BLE001
        return 0


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two physical GPUs (RCCL refuses two ranks on one device)")
@pytest.mark.parametrize("exchange", ["rccl", "p2p"])
@pytest.mark.parametrize("gens", [9, 130])
def test_two_physical_gpus_rccl_and_peer_exchange(exchange, gens):
    """The first box with more than one GPU runs this: one rank per device, process group nccl (RCCL over xGMI), the
    record exchange through RCCL all-gathers ("rccl": captured into a graph from 50 generations on) and through peer
    writes into IPC-mapped HBM across a real link ("p2p", after its probe) -- == oracle.run_de_sharded bit for bit on
    every rank.  Everywhere else these transports are only ever exercised with ranks sharing one device."""
    from _dist_workers import nccl_multi_gpu_worker

    world = min(_n_gpus(), 8)
    world = 1 << (world.bit_length() - 1)  # 2, 4 or 8
    _check_sharded_de(world, _de_cfg(24, 64 * world, gens, 2024 + gens, exchange), worker=nccl_multi_gpu_worker)


@pytest.mark.gpu
@pytest.mark.skipif(_n_gpus() < 2, reason="needs two physical GPUs")
@pytest.mark.parametrize("method", ["pso", "cpso"])
def test_two_physical_gpus_pso_is_exact(method):
    """PSO / CPSO sharded over physical GPUs (nccl process group; best record by peer writes or all-gather, the
    restart's [pbestfit | radii] gather over RCCL) == the unsharded oracle run."""
    from _dist_workers import nccl_multi_gpu_worker

    world = 2
    opts = {"maxiter": 70, "popsize": 256, "seed": 5, "ftol": -1.0, "xtol": 0.0}
    cfg = {"n": 16, "objective": "sphere", "method": method, "options": opts}
    out = _spawn(nccl_multi_gpu_worker, world, cfg)
    ref = oracle.minimize("sphere", [[-5.12, 5.12]] * 16, method=method, options=dict(opts, updating="deferred"), rng="philox")
    for r in range(world):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)


@pytest.mark.gpu
@pytest.mark.parametrize("method,n,P,world", [("pso", 20, 97, 2), ("pso", 20, 130, 4), ("cmaes", 20, 49, 2), ("vdcma", 40, 27, 2)])
def test_any_popsize_is_sharded_the_last_rank_short(method, n, P, world):
    """Round 6 (VERDICT r5 next #6b): popsize need not be a multiple of workers -- blocks of ceil(P / workers) rows, the last
    rank takes what is left (parallel.shard_bounds; the reference's MPI loop takes any popsize, _common.py:64-65).  PSO is
    row-local, CMA-ES / VD-CMA gather padded blocks: the sharded run is the unsharded one on every rank."""
    import stochopy_amd as sa
    from _dist_workers import gpu_minimize_worker

    opts = {"maxiter": 25, "popsize": P, "seed": 21, "ftol": -1.0, "xtol": 0.0}
    if method != "pso":
        opts["rng"] = "philox"
    cfg = {"n": n, "objective": "rosenbrock", "method": method, "options": opts, "rng": "philox"}
    out = _spawn(gpu_minimize_worker, world, cfg)
    if method == "pso":
        ref = oracle.minimize("rosenbrock", [[-5.12, 5.12]] * n, method="pso", options=dict(opts, updating="deferred"), rng="philox")
    else:  # (the sharded device loop runs the one-GPU resident loop's kernels: bit-identical to it)
        ref = sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method=method, options=dict(opts, backend="hip"))
    for r in range(world):
        fun, nit, nfev, status = np.load(os.path.join(out, f"meta_{r}.npy"))
        assert (fun, nit, nfev, status) == (ref.fun, ref.nit, ref.nfev, ref.status)
        assert np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref.x)


def test_cpso_restart_needs_equal_shards_and_says_so():
    """The competitive restart's swarm-wide selection gathers equal segments: an uneven swarm is refused by name, before any
    collective (every rank raises alike); shard_bounds itself refuses only populations that leave the last rank nothing."""
    from stochopy_amd import parallel

    assert parallel.shard_bounds(97, 2, 1) == (49, 48)
    with pytest.raises(ValueError, match="leave the last rank"):
        parallel.shard_bounds(9, 8, 0)
