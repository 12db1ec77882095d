"""Worker entry points for the multi-process tests (spawned; must be importable)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _init(rank, world, port):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def cpu_exchange_worker(rank, world, port, cfg, out_dir):
    """CPU-only: each rank advances ITS shard with the oracle's arithmetic and exchanges the global best
    through stochopy_amd.parallel.World over gloo -- the product's exchange code, real processes."""
    import torch

    dist = _init(rank, world, port)
    try:
        import oracle
        from oracle import engine as oe
        from stochopy_amd import parallel

        w = parallel.require_world(world)
        n, P, gens = cfg["n"], cfg["P"], cfg["gens"]
        lower, upper = np.full(n, -5.12), np.full(n, 5.12)
        stream = oracle.PhiloxStream(cfg["seed"])
        row0, Pl = w.shard(P)
        fobj = oracle.OBJECTIVES["rosenbrock"]
        X = oe.latin_hypercube(stream, P, n, lower, upper)[row0 : row0 + Pl].copy()
        fit = fobj(X)

        def exchange():
            g = int(np.argmin(fit))
            rec = torch.from_numpy(np.concatenate([[fit[g], float(row0 + g)], X[g]]))
            out = torch.empty((world, n + 2), dtype=torch.float64)
            w.all_gather_records(rec, out)
            wb, f, gi = parallel.best_of_records(out.numpy())
            return out[wb, 2:].numpy().copy(), f, gi

        gbest, gfit, _ = exchange()
        trace = [gfit]
        for it in range(2, gens + 1):
            draws = stream.de_generation(it, Pl, n, 2, None, row0=row0)
            U = oe.de_candidates(X, gbest, draws, 0.5, 0.9, "best1bin", lower, upper, None)
            cf = fobj(U)
            better = cf < fit
            fit[better] = cf[better]
            X[better] = U[better]
            gbest, gfit, _ = exchange()
            trace.append(gfit)
        assert w.max_over_ranks(rank) == world - 1
        np.save(os.path.join(out_dir, f"trace_{rank}.npy"), np.array(trace))
    finally:
        dist.destroy_process_group()


def cpu_rows_worker(rank, world, port, cfg, out_dir):
    """CPU-only: World.all_gather_rows / all_gather_object / all_agree / barrier over gloo."""
    import torch

    dist = _init(rank, world, port)
    try:
        from stochopy_amd import parallel

        w = parallel.require_world(world)
        n = cfg["n"]
        total = cfg["total"] if "total" in cfg else world * cfg["rows"]
        rows = w.shard(total)[1]  # (blocks of ceil(total / world) rows, the last rank short)
        local = (torch.arange(w.shard_rows(total) * n, dtype=torch.float64).reshape(-1, n) + 1000.0 * rank)[:rows].contiguous()
        out = torch.empty((total, n), dtype=torch.float64)
        w.all_gather_rows(local, out)
        fit = torch.empty((total,), dtype=torch.float64)
        w.all_gather_rows(local[:, 0].contiguous(), fit)
        objs = w.all_gather_object({"rank": rank})
        assert [o["rank"] for o in objs] == list(range(world))
        assert w.all_agree(True) and not w.all_agree(rank != 1)
        w.barrier()
        np.save(os.path.join(out_dir, f"rows_{rank}.npy"), out.numpy())
        np.save(os.path.join(out_dir, f"fit_{rank}.npy"), fit.numpy())
    finally:
        dist.destroy_process_group()


def cpu_cpso_worker(rank, world, port, cfg, out_dir):
    """CPU-only: competitive PSO with the swarm sharded by rows.  Every rank moves ITS particles with the oracle's
    arithmetic (Philox draws keyed by the global row) and the two swarm-wide steps go through the product's exchange
    code with the product's buffer layouts: the best record (World.all_gather_records of [f, global row, x]) and the
    competitive restart's ONE gather of [pbestfit | per-part radii] (optimize/_cpso.py `fit_radius` ->
    `fit_radius_all`), from which every rank derives the same radius and the same worst-nw rows."""
    import torch

    dist = _init(rank, world, port)
    try:
        import oracle
        from oracle import engine as oe
        from stochopy_amd import parallel

        w = parallel.require_world(world)
        n, P, maxiter, gamma, npart = cfg["n"], cfg["P"], cfg["maxiter"], cfg["gamma"], cfg["npart"]
        lower, upper = np.full(n, -32.768), np.full(n, 32.768)
        stream = oracle.PhiloxStream(cfg["seed"])
        fobj = oracle.OBJECTIVES[cfg["objective"]]
        row0, Pl = w.shard(P)
        delta = np.log(1.0 + 0.003 * P) / np.max((0.2, np.log(0.01 * maxiter)))
        X = oe.latin_hypercube(stream, P, n, lower, upper)[row0:row0 + Pl].copy()
        V = np.zeros((Pl, n))
        pbest, pbestfit = X.copy(), fobj(X)

        def exchange_best():
            g = int(np.argmin(pbestfit))
            rec = torch.from_numpy(np.concatenate([[pbestfit[g], float(row0 + g)], pbest[g]]))
            out = torch.empty((world, n + 2), dtype=torch.float64)
            w.all_gather_records(rec, out)
            wb, f, gi = parallel.best_of_records(out.numpy())
            return out[wb, 2:].numpy().copy(), f

        gbest, gfit = exchange_best()
        it, restarts, rows_log, trace = 1, [], [], [gfit]
        while True:
            it += 1
            r1, r2 = stream.pso_generation(it, Pl, n, row0=row0)
            X, V = oe.pso_move(X, V, pbest, gbest, 0.7298, 1.49618, 1.49618, r1, r2, lower, upper, cfg["constraints"])
            pfit = fobj(X)
            better = pfit < pbestfit
            pbest[better], pbestfit[better] = X[better], pfit[better]
            new_best, new_fit = exchange_best()
            # (engine.greedy_select / termination with the gathered best)
            status = None
            if new_fit < gfit:
                dx = np.linalg.norm(new_best - gbest)
                gbest, gfit = new_best, new_fit
                status = oe.termination(it, maxiter, dx, gfit, cfg["xtol"], cfg["ftol"])
            elif it >= maxiter:
                status = -1
            trace.append(gfit)
            if status is not None:
                break
            # the restart: per-part radii of the shard (parts of consecutive rows, as the device's workgroups leave them)
            d = X - gbest
            rr = np.sqrt((d * d).sum(axis=1))
            parts = np.array([rr[k::npart].max() if len(rr[k::npart]) else 0.0 for k in range(npart)])
            fit_radius = torch.from_numpy(np.concatenate([pbestfit, parts]))
            fit_radius_all = torch.empty((world, Pl + npart), dtype=torch.float64)
            w.all_gather_records(fit_radius, fit_radius_all)
            allv = fit_radius_all.numpy()
            radius = allv[:, Pl:].max() / np.sqrt(4.0 * n)
            if radius < delta:
                nw = oe.restart_count(it, maxiter, P, gamma)
                if nw > 0:
                    rows = allv[:, :Pl].reshape(-1).argsort()[: -nw - 1 : -1]  # global rows, rank-major = row order
                    mine = np.sort(rows[(rows >= row0) & (rows < row0 + Pl)]) - row0
                    V[mine] = 0.0
                    X[mine] = stream.restart_rows(it, lower, upper, mine, n, row0=row0)
                    pbest[mine] = X[mine]
                    pbestfit[mine] = 1.0e30
                    restarts.append((it, nw))
                    rows_log.append(np.sort(rows))
        np.savez(os.path.join(out_dir, f"cpso_{rank}.npz"), x=gbest, fun=gfit, nit=it, status=status, trace=np.array(trace),
                 restarts=np.array(restarts).reshape(-1, 2), rows=np.concatenate(rows_log) if rows_log else np.zeros(0))
    finally:
        dist.destroy_process_group()


def cpu_cma_worker(rank, world, port, cfg, out_dir):
    """CPU-only: CMA-ES (and its Penalize form) shards what the reference's parallel backends shard -- the candidates.
    Every rank draws the normals of ITS rows (keyed by the global row) and evaluates ITS candidates; the product's
    World.all_gather_rows returns all rows / fitness values in rank order and the model update is replicated.  The run
    must be the unsharded oracle run bit for bit on every rank."""
    import torch

    dist = _init(rank, world, port)
    try:
        import oracle
        from oracle import engine as oe
        from stochopy_amd import parallel

        w = parallel.require_world(world)
        n, P = cfg["n"], cfg["P"]
        lower, upper = np.full(n, -3.0), np.full(n, 3.0)
        fobj = oracle.OBJECTIVES[cfg["objective"]]
        row0, Pl = w.shard(P)
        shard0 = row0

        class ShardedStream(oracle.PhiloxStream):
            def cma_normals(self, gen, P_, n_, row0=0):
                if P_ != P:  # (VD-CMA's injection normals: one replicated row keyed past the population)
                    return super().cma_normals(gen, P_, n_, row0=row0)
                loc = torch.from_numpy(np.ascontiguousarray(super().cma_normals(gen, Pl, n_, row0=shard0)))
                out = torch.empty((P_, n_), dtype=torch.float64)
                w.all_gather_rows(loc, out)
                return out.numpy().copy()

        calls = []

        def sharded_fobj(Xfull):
            loc = torch.from_numpy(np.ascontiguousarray(fobj(Xfull[row0:row0 + Pl])))
            out = torch.empty((len(Xfull),), dtype=torch.float64)
            w.all_gather_rows(loc, out)
            calls.append(len(Xfull))
            return out.numpy().copy()

        method = oe.run_vdcma if cfg.get("method") == "vdcma" else oe.run_cmaes
        res = method(sharded_fobj, lower, upper, None, ShardedStream(cfg["seed"]), maxiter=cfg["maxiter"], popsize=P,
                     sigma=0.3, constraints=cfg.get("constraints"), eigh="canonical")
        np.savez(os.path.join(out_dir, f"cma_{rank}.npz"), x=res["x"], fun=res["fun"], nit=res["nit"], status=res["status"],
                 calls=len(calls))
    finally:
        dist.destroy_process_group()


def _spy_on_de_runs():
    """Record the _DeRun / _PsoRun objects minimize() creates (to see which exchange a run ended up with)."""
    from stochopy_amd.optimize import _cpso, _de

    runs = []
    for cls in (_de._DeRun, _cpso._PsoRun):
        orig = cls.__init__

        def spy(self, *a, _orig=orig, **k):
            runs.append(self)
            _orig(self, *a, **k)

        cls.__init__ = spy
    return runs


def _minimize_and_save(rank, world, cfg, out_dir):
    import stochopy_amd as sa

    os.environ.update(cfg.get("env", {}))
    runs = _spy_on_de_runs()
    n = cfg["n"]
    opts = dict(cfg["options"], backend="hip", workers=world)
    opts.setdefault("rng", "philox")
    fun = getattr(sa.factory, cfg["objective"])
    if cfg.get("external") == "batched":  # a caller-supplied device objective (torch ops on the shard's rows)
        fun = sa.factory.batched(lambda X: (X * X).sum(dim=1))
    elif cfg.get("external") == "device-sphere":  # the fused kernel's own values, handed in as a caller's objective
        import ctypes as C

        import torch

        from stochopy_amd import _lib

        def _sphere(X):
            f = torch.empty((X.shape[0],), dtype=torch.float64, device=X.device)
            Xc = X.contiguous()
            assert _lib.lib().sx_eval(_lib.FUN_IDS["sphere"], Xc.data_ptr(), X.shape[0], X.shape[1], X.shape[1], None, None,
                                      f.data_ptr(), None, None, C.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0
            return f

        fun = sa.factory.batched(_sphere)
    seen = []
    cb = (lambda X, r: seen.append((np.array(X, copy=True), float(r.fun), int(r.nit), int(r.nfev)))) if cfg.get("callback") else None
    res = sa.optimize.minimize(fun, cfg.get("bounds", [[-5.12, 5.12]] * n), method=cfg["method"], options=opts, callback=cb)
    if "xall" in res:
        np.save(os.path.join(out_dir, f"xall_{rank}.npy"), res.xall)
        np.save(os.path.join(out_dir, f"funall_{rank}.npy"), res.funall)
    if cb is not None:
        np.save(os.path.join(out_dir, f"cbX_{rank}.npy"), np.array([c[0] for c in seen]))
        np.save(os.path.join(out_dir, f"cbmeta_{rank}.npy"), np.array([c[1:] for c in seen]))
    np.save(os.path.join(out_dir, f"x_{rank}.npy"), res.x)
    np.save(os.path.join(out_dir, f"meta_{rank}.npy"), np.array([res.fun, res.nit, res.nfev, res.status]))
    if runs:
        with open(os.path.join(out_dir, f"exchange_{rank}.txt"), "w") as f:
            f.write(str(runs[-1].exchange))


def gpu_minimize_worker(rank, world, port, cfg, out_dir):
    """Ranks share the one GPU of the test box (gloo for the process group): the sharded HIP path end to end."""
    dist = _init(rank, world, port)
    try:
        import torch

        torch.cuda.set_device(0)
        _minimize_and_save(rank, world, cfg, out_dir)
    finally:
        dist.destroy_process_group()


def nccl_multi_gpu_worker(rank, world, port, cfg, out_dir):
    """One rank per PHYSICAL GPU, backend nccl (= RCCL over xGMI): the deployment.  Needs >= world devices (the test
    that uses it is skipped on one-GPU boxes)."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = os.environ["TORCH_FR_BUFFER_SIZE"] = "256"
    torch.cuda.set_device(rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        _minimize_and_save(rank, world, cfg, out_dir)
    finally:
        dist.destroy_process_group()


def gpu_late_failure_worker(rank, world, port, cfg, out_dir):
    """Peer exchange, callback on every rank; rank 1's callback raises at the LAST generation, i.e. after its peers' final
    kernels are queued: every rank must come out of minimize() with an exception (rank 1 its own, the others "a peer
    rank failed"), nobody may be left in a collective the failing rank never enters."""
    dist = _init(rank, world, port)
    try:
        import torch

        torch.cuda.set_device(0)
        import stochopy_amd as sa

        os.environ.update(cfg.get("env", {}))
        n, opts = cfg["n"], dict(cfg["options"], backend="hip", workers=world, rng="philox")
        last = opts["maxiter"]

        def cb(X, r):
            if rank == 1 and int(r.nit) >= last:
                raise ValueError("callback failed on purpose")

        try:
            sa.optimize.minimize(getattr(sa.factory, cfg["objective"]), [[-5.12, 5.12]] * n, method=cfg["method"],
                                 options=opts, callback=cb)
            msg = "no error"
        except Exception as e:  # noqa: BLE001
            msg = f"{type(e).__name__}: {e}"
        with open(os.path.join(out_dir, f"err_{rank}.txt"), "w") as f:
            f.write(msg)
        dist.barrier()  # both ranks are out of minimize() and their process group still works
    finally:
        dist.destroy_process_group()


def gpu_p2p_straggler_worker(rank, world, port, cfg, out_dir):
    """Rank 1 sets the exchange up and then never launches a generation; rank 0 must time out and raise."""
    import time

    os.environ["SX_XCHG_TIMEOUT_S"] = "1"
    dist = _init(rank, world, port)
    try:
        import torch

        torch.cuda.set_device(0)
        import stochopy_amd as sa
        from stochopy_amd import _lib
        from stochopy_amd.optimize import _de

        n, o = cfg["n"], cfg["options"]
        run = _de._DeRun(_lib.FUN_IDS[cfg["objective"]], np.full(n, -5.12), np.full(n, 5.12), None, o["maxiter"],
                         o["popsize"], 0.5, 0.9, "best1bin", None, 0.0, -1.0, False, 1.0, None, "philox", o["seed"],
                         world, autorun=False, exchange="p2p")
        assert run.exchange == "p2p"
        with torch.cuda.stream(run.ctx.stream):
            run._setup()
            if rank == 0:
                run.enqueue(3)
                try:
                    run.read_state()
                    msg = "no error"
                except RuntimeError as e:
                    msg = str(e)
                with open(os.path.join(out_dir, "err_0.txt"), "w") as f:
                    f.write(msg)
            else:
                time.sleep(0.5)
        dist.barrier()
        run.close()
    finally:
        dist.destroy_process_group()


def nccl_single_rank_worker(rank, world, port, cfg, out_dir):
    """One rank, backend nccl (= RCCL): the production exchange path (all_gather_into_tensor on the device)."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SX_FORCE_SHARDED"] = "1"
    # (must be set before the group exists; with it graph captures wait on the flight recorder, without it they sleep:
    #  parallel.World.quiesce_for_capture -- the tests cover both)
    if cfg.get("flight_recorder"):
        os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = os.environ["TORCH_FR_BUFFER_SIZE"] = "256"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        _minimize_and_save(rank, 1, cfg, out_dir)
    finally:
        dist.destroy_process_group()


def nccl_single_rank_blocks_worker(rank, world, port, cfg, out_dir):
    """One rank over RCCL, the run stepped in blocks of K < GRAPH_CHUNK generations the way `bench.py --steps 20` drives it:
    from the second block on the block is one replay of a captured K-generation graph (kernels + all-gathers)."""
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SX_FORCE_SHARDED"] = "1"
    os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = os.environ["TORCH_FR_BUFFER_SIZE"] = "256"
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    try:
        from stochopy_amd import _lib
        from stochopy_amd.optimize import _de

        n, P, K, blocks = cfg["n"], cfg["P"], cfg["K"], cfg["blocks"]
        out = {}
        for label, steps in (("blocks", [K] * blocks), ("once", [K * blocks])):
            run = _de._DeRun(_lib.FUN_IDS["rosenbrock"], np.full(n, -5.12), np.full(n, 5.12), None, 2**31 - 2, P, 0.5, 0.9,
                             "best1bin", None, 0.0, -1.0, False, 1.0, None, "philox", cfg["seed"], 1, autorun=False, exchange="rccl")
            try:
                with torch.cuda.stream(run.ctx.stream):
                    run._setup()
                    for k in steps:
                        run.enqueue(k)
                    run.ctx.sync()
                    st = run.read_state()
                    out[label] = (int(st.it), float(st.gfit), sorted(run._rccl_graphs), run._rccl_graph_note)
                    out[label + "_x"] = run.bufs[st.it & 1].cpu().numpy().copy()
            finally:
                run.close()
        assert out["blocks"][0] == out["once"][0] == 1 + K * blocks, out
        assert out["blocks"][1] == out["once"][1] and np.array_equal(out["blocks_x"], out["once_x"])
        assert out["blocks"][3] is None and out["blocks"][2] == [K], out["blocks"]  # the K-generation graph was captured and used
        np.save(os.path.join(out_dir, f"blocks_{rank}.npy"), np.array([out["blocks"][0], out["blocks"][1]]))
    finally:
        dist.destroy_process_group()
