"""Long-run stability of the peer exchange: 4 ranks on one GPU, many generations, p2p vs rccl(gloo) transports
must agree bit for bit (DE shard-local donors, DE global donors vs the unsharded single-GPU run, PSO)."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/tests")


def main():
    import torch.multiprocessing as mp
    from _dist_workers import gpu_minimize_worker
    from test_distributed import _free_port

    gens = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    world, n, P = 4, 64, 1024
    results = {}
    for name, method, opts, env in (
        ("de_p2p", "de", {"exchange": "p2p"}, {}),
        ("de_rccl", "de", {"exchange": "rccl"}, {}),
        ("de_global", "de", {"exchange": "p2p", "donors": "global"}, {}),
        ("pso_p2p", "pso", {}, {"SX_EXCHANGE": "p2p"}),
        ("pso_rccl", "pso", {}, {"SX_EXCHANGE": "rccl"}),
    ):
        o = {"maxiter": gens, "popsize": P, "seed": 123, "ftol": -1.0, "xtol": 0.0, "updating": "deferred"}
        o.update(opts)
        cfg = {"n": n, "objective": "rastrigin", "method": method, "options": o, "env": env}
        out = tempfile.mkdtemp(prefix="sx_stress_")
        mp.spawn(gpu_minimize_worker, args=(world, _free_port(), cfg, out), nprocs=world, join=True)
        metas = [np.load(os.path.join(out, f"meta_{r}.npy")) for r in range(world)]
        xs = [np.load(os.path.join(out, f"x_{r}.npy")) for r in range(world)]
        assert all(np.array_equal(metas[0], m) for m in metas) and all(np.array_equal(xs[0], x) for x in xs), name
        results[name] = (metas[0], xs[0])
        print(name, "fun %.17g nit %d" % (metas[0][0], metas[0][1]), flush=True)
    assert np.array_equal(results["de_p2p"][0], results["de_rccl"][0]) and np.array_equal(results["de_p2p"][1], results["de_rccl"][1])
    assert np.array_equal(results["pso_p2p"][0], results["pso_rccl"][0]) and np.array_equal(results["pso_p2p"][1], results["pso_rccl"][1])
    import stochopy_amd as sa

    one = sa.optimize.minimize(sa.factory.rastrigin, [[-5.12, 5.12]] * n, method="de",
                               options={"maxiter": gens, "popsize": P, "seed": 123, "ftol": -1.0, "xtol": 0.0,
                                        "rng": "philox", "updating": "deferred"})
    assert one.fun == results["de_global"][0][0] and np.array_equal(one.x, results["de_global"][1])
    print("stress ok: transports agree over %d generations; global donors == single GPU" % gens)


if __name__ == "__main__":
    main()
