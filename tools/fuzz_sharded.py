"""Randomised multi-rank runs on ONE GPU (gloo for set-up, both transports): sharded PSO / CPSO and DE with global
donors must equal the unsharded oracle run, DE with shard-local donors the sharded oracle.  usage: fuzz_sharded.py [cases] [seed]"""
import os, sys, tempfile, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np


def main():
    import torch.multiprocessing as mp
    import oracle
    from oracle import engine as oe
    from _dist_workers import gpu_minimize_worker
    from test_distributed import _free_port

    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    bad, t0 = 0, time.time()
    for c in range(cases):
        world = int(rs.choice([2, 2, 3, 4, 8]))
        method = str(rs.choice(["de", "de", "pso", "cpso"]))
        n = int(rs.choice([2, 8, 24, 64, 100, 128, 200, 300]))
        P = world * int(rs.randint(6, 80))
        gens = int(rs.randint(3, 60))
        exchange = str(rs.choice(["p2p", "rccl"]))
        objective = str(rs.choice(["sphere", "rosenbrock"]))
        o = {"maxiter": gens, "popsize": P, "seed": int(rs.randint(1 << 30)), "updating": "deferred"}
        if rs.rand() < 0.3:
            o.update(ftol=float(10 ** rs.uniform(-2, 3)), xtol=float(10 ** rs.uniform(-3, 1)))
        else:
            o.update(ftol=-1.0, xtol=0.0)
        env, glob = {}, False
        if method == "de":
            o["strategy"] = str(rs.choice(["rand1bin", "rand2bin", "best1bin", "best2bin"]))
            if rs.rand() < 0.3:
                o["constraints"] = "Random"
                o["mutation"] = float(rs.uniform(0.8, 1.6))
            o["exchange"] = exchange
            glob = exchange == "p2p" and rs.rand() < 0.5
            if glob:
                o["donors"] = "global"
        else:
            env["SX_EXCHANGE"] = exchange
            if rs.rand() < 0.5:
                o.update(constraints="Shrink", inertia=0.91)
        cfg = {"n": n, "objective": objective, "method": method, "options": o, "env": env}
        out = tempfile.mkdtemp(prefix="sx_fuzz_")
        mp.spawn(gpu_minimize_worker, args=(world, _free_port(), cfg, out), nprocs=world, join=True)
        oo = {k: v for k, v in o.items() if k not in ("exchange", "donors")}
        b = [[-5.12, 5.12]] * n
        if method == "de" and not glob:
            ref = oe.run_de_sharded(oracle.OBJECTIVES[objective], np.full(n, -5.12), np.full(n, 5.12),
                                    oracle.PhiloxStream(o["seed"]), world, **{k: v for k, v in oo.items() if k != "seed"})
        else:
            ref = oracle.minimize(objective, b, method=method, options=dict(oo), rng="philox")
        ok = True
        for r in range(world):
            meta = np.load(os.path.join(out, f"meta_{r}.npy"))
            ok = ok and tuple(meta) == (ref["fun"], ref["nit"], ref["nfev"], ref["status"])
            ok = ok and np.array_equal(np.load(os.path.join(out, f"x_{r}.npy")), ref["x"])
        if not ok:
            bad += 1
            print("MISMATCH", c, world, method, n, P, gens, o, env, flush=True)
    print(f"{cases} sharded cases, {bad} mismatches, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
