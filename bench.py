#!/usr/bin/env python3
"""Benchmark of the per-generation hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.
A "step" is one DE generation (donor gather, mutation, crossover, objective,
selection, best/termination) over the whole resident population.

Workload at N=1 (BASELINE.json metric): DE best1bin, Rosenbrock, dim=128,
popsize=4096, F=0.5, CR=0.9, bounds +-5.12, in-kernel Philox draws, ftol=-1 /
xtol=0 so every step does the full work.  N>1: the same shard per GPU (weak
scaling), one process per GPU, global best exchanged every generation.

`value` = objective evaluations per second = N * popsize * K / t(K steps),
population resident in HBM before the timed region.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s measured float4 copy)

WORKLOADS = {
    # name: (objective, n, P, strategy)
    "de_rosenbrock_n128_p4096": ("rosenbrock", 128, 4096, "best1bin"),
    "de_rastrigin_n128_p4096": ("rastrigin", 128, 4096, "best1bin"),
    "de_rosenbrock_n1024_p16384": ("rosenbrock", 1024, 16384, "best1bin"),
    "de_rosenbrock_n1024_p131072": ("rosenbrock", 1024, 131072, "best1bin"),
    "de_rosenbrock_n256_p4096": ("rosenbrock", 256, 4096, "best1bin"),
    "de_rosenbrock_n512_p8192": ("rosenbrock", 512, 8192, "best1bin"),
    "de_rastrigin_n1024_p16384": ("rastrigin", 1024, 16384, "best1bin"),
    "de_rosenbrock_n2048_p16384": ("rosenbrock", 2048, 16384, "best1bin"),
    "de_rand1bin_n1024_p16384": ("rosenbrock", 1024, 16384, "rand1bin"),
}


def algorithmic_bytes_per_eval(n, k):
    """read X_i + k donor rows, write the candidate row, r/w fitness."""
    return 8 * n * (k + 2) + 16


def cpu_baseline(objective, n, P, strategy, budget_s=12.0):
    """The oracle (numpy port of the reference loop, numpy-legacy stream) on ONE host core."""
    import oracle

    gens = []

    def cb(X, r):
        gens.append(time.perf_counter())
        if len(gens) >= 3 and gens[-1] - gens[0] > budget_s:
            raise StopIteration

    t0 = time.perf_counter()
    try:
        oracle.minimize(objective, [[-5.12, 5.12]] * n, method="de", callback=cb,
                        options={"maxiter": 10**6, "popsize": P, "seed": 0, "strategy": strategy, "ftol": -1.0,
                                 "xtol": 0.0})
    except StopIteration:
        pass
    done = len(gens) - 1  # generations after the initial evaluation
    dt = gens[-1] - gens[0]
    return {
        "value": P * done / dt,
        "unit": "evals/s",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle DE {strategy} {objective} n={n} P={P}, {done} generations in {dt:.1f}s, serial numpy "
                  f"(numpy-legacy stream incl. the reference's O(P^2) donor permutations), "
                  f"host cpu_count={os.cpu_count()}",
        "t_first_eval_s": gens[0] - t0,
    }


def _one_row(objective, x):
    import oracle

    return float(oracle.OBJECTIVES[objective](x[None, :])[0])


def cpu_baseline_loky(objective, n, P, strategy, budget_s=6.0):
    """The same oracle loop with the objective farmed out the way the reference's joblib backend does it
    (_common.py:39-43: one task per individual, `Parallel(n_jobs=workers)(delayed(fun)(x) for x in X)`), on all
    host cores through joblib's loky pool (kept warm across generations).  Reported next to the serial figure:
    for objectives this cheap the per-task overhead makes it SLOWER than serial, as with the reference itself."""
    import oracle
    from joblib import Parallel, delayed

    cores = min(os.cpu_count() or 1, 32)  # (pool start-up grows with the worker count; 32 is past the point of any gain)
    stamps = []
    with Parallel(n_jobs=cores, backend="loky") as pool:
        def fobj(X):
            return np.array(pool(delayed(_one_row)(objective, x) for x in X))

        def cb(X, r):
            stamps.append(time.perf_counter())
            if len(stamps) >= 3 and stamps[-1] - stamps[0] > budget_s:
                raise StopIteration

        try:
            oracle.minimize(fobj, [[-5.12, 5.12]] * n, method="de", callback=cb,
                            options={"maxiter": 10**6, "popsize": P, "seed": 0, "strategy": strategy, "ftol": -1.0,
                                     "xtol": 0.0})
        except StopIteration:
            pass
    done, dt = len(stamps) - 1, stamps[-1] - stamps[0]
    return {"value": P * done / dt, "unit": "evals/s", "cores": cores, "kind": "port",
            "sample": f"oracle DE {strategy} {objective} n={n} P={P}, {done} generations in {dt:.1f}s, one joblib-loky task "
                      f"per individual on {cores} workers (the reference's parallel backend scheme), host cpu_count={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="de_rosenbrock_n128_p4096", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="CPU work spent on the baseline sample")
    ap.add_argument("--kernel-timing-launches", type=int, default=400)
    args = ap.parse_args()

    import torch

    import stochopy_amd as sa  # noqa: F401  (fails loudly here if the HIP library is missing)
    from stochopy_amd import _lib
    from stochopy_amd.optimize import _de

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or ("RANK" in os.environ and os.environ.get("SX_FORCE_SHARDED") == "1"):
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    objective, n, P, strategy = WORKLOADS[args.workload]
    k = _lib.DE_DONORS[strategy]
    K, W = args.steps, args.warmup
    lower = np.full(n, -5.12)
    upper = np.full(n, 5.12)
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(exchange):
        # weak scaling: every GPU owns P rows of a global population of world*P (same seed on every rank)
        run = _de._DeRun(_lib.FUN_IDS[objective], lower, upper, None, 2**31 - 2, world * P, 0.5, 0.9, strategy, None,
                         0.0, -1.0, False, 1.0, None, "philox", 1234, world, autorun=False, exchange=exchange,
                         donors=os.environ.get("SX_DONORS"))
        ctx = run.ctx
        try:
            with torch.cuda.stream(ctx.stream):
                run._setup()
                run.prepare_graphs()
                run.enqueue(W)
                ctx.sync()
                barrier()
                t0 = time.perf_counter()
                run.enqueue(K)
                ctx.sync()
                barrier()
                t1 = time.perf_counter()
                st = run.read_state()  # raises if a wait inside the peer exchange timed out
                assert st.it == 1 + W + K, (st.it, W, K)

                # dominant kernel: HIP events on the engine stream around a replayed hipGraph of generation
                # kernels (real generations, nothing else on the stream), average per launch
                nl = (args.kernel_timing_launches // run.GRAPH_CHUNK) * run.GRAPH_CHUNK
                ev0 = torch.cuda.Event(enable_timing=True)
                ev1 = torch.cuda.Event(enable_timing=True)
                run.enqueue(run.GRAPH_CHUNK)
                ev0.record(ctx.stream)
                run.enqueue(nl)
                ev1.record(ctx.stream)
                ctx.sync()
                kernels_per_gen = 1 if run.chain else (2 if run.world is None else 3)
                kern_ms = ev0.elapsed_time(ev1) / nl
            barrier()  # no rank frees its exchange buffer while a peer may still write into it
        finally:
            run.close()
        return run, t1 - t0, kern_ms, nl, kernels_per_gen

    try:
        run, dt, kern_ms, nl, kernels_per_gen = measure(None)
    except RuntimeError as e:
        # a peer-exchange wait that timed out mid-run (every rank sees it within one timeout): the transport
        # passed its self-test but is not usable here -- measure through the RCCL transport instead and say so
        if dist is None or "peer exchange" not in str(e):
            raise
        print(f"[bench] rank {rank}: {e}; falling back to exchange='rccl'", file=sys.stderr, flush=True)
        run, dt, kern_ms, nl, kernels_per_gen = measure("rccl")
        run.exchange_note = f"p2p failed mid-run ({e})"

    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    value = world * P * K / dt

    if rank == 0:
        bytes_per_launch = algorithmic_bytes_per_eval(n, k) * P
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:
                traffic = json.load(open(pmc)).get(args.workload, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        line = {
            "metric": "objective-fn evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": args.workload,
                "method": "de", "strategy": strategy, "objective": objective, "dim": n,
                "popsize_per_gpu": P, "popsize_total": world * P, "rng": "philox", "F": 0.5, "CR": 0.9,
                "exchange": ("none" if run.world is None else
                             "global best per generation, peer writes over xGMI inside the generation kernel "
                             "(tagged (n+2)-double record per rank)" if run.exchange == "p2p" else
                             "global best per generation (RCCL all_gather of an (n+2)-double record)"),
                "exchange_note": getattr(run, "exchange_note", None),
                "donors": (None if run.world is None else "global (rows read from their owners over xGMI)"
                           if run.global_donors else "shard-local (island model with a shared global best)"),
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "de_generation_kernel<%s,philox>" % objective,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel_us": kern_ms * 1e3,
                "timing": "HIP events on the engine stream around %d generations (%d kernel(s) each)" % (nl, kernels_per_gen),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(objective, n, min(P, 4096), strategy, args.cpu_baseline_seconds)
            line["gpu_over_cpu"] = value / line["cpu_baseline"]["value"]
            try:  # the reference's own parallel-backend scheme, as a second reported baseline
                line["cpu_baseline_loky"] = cpu_baseline_loky(objective, n, min(P, 4096), strategy,
                                                             min(6.0, args.cpu_baseline_seconds))
            except Exception as e:  # noqa: BLE001  (joblib missing / pool failure: say so, keep the line)
                line["cpu_baseline_loky"] = {"error": str(e)}
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
