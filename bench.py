#!/usr/bin/env python3
"""Benchmark of the per-generation hot path on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.
A "step" is one DE generation (donor gather, mutation, crossover, objective,
selection, best/termination) over the whole resident population.

Workload at N=1 (BASELINE.json metric): DE best1bin, Rosenbrock, dim=128,
popsize=4096, F=0.5, CR=0.9, bounds +-5.12, in-kernel Philox draws, ftol=-1 /
xtol=0 so every step does the full work.  N>1: the same shard per GPU (weak
scaling), one process per GPU, global best exchanged every generation.

`value` = objective evaluations per second = N * popsize * (steps timed) / t, population resident in HBM
before the timed region.  The K-step block is repeated back to back until the timed region lasts >= 2 s
(`blocks`; K steps alone take 0.15 ms at the driver's K = 20), bracketed by barrier + synchronize; the median block
duration (HIP events between blocks) is reported next to it.  Also in the line:
`minimize_wall` -- evals/s of a whole `minimize()` call after one warm-up call (host-side initial population
included); `cpu_baseline` / `cpu_baseline_loky` with core count and CPU model; for N > 1 `c5` -- BASELINE config 5
(DE n=1024, P=131072 in total = strong scaling) through both donor modes, with the number of ranks seen.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# Host BLAS on ONE thread (set before numpy loads).  The CPU baseline is quoted as "cores: 1", and on the GPU boxes the container
# has a CPU quota of 16 on a 256-CPU host (cgroup cpu.max 1600000 100000) while OpenBLAS starts 64 spinning threads for any
# vector of more than ~10^4 elements: a single np.dot of 16 384 elements got the whole process throttled for tens of
# milliseconds (profiles/r5_host_blas_throttle.txt) -- VD-CMA's set-up does one, and the "configs" timings are wall clock.
os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
os.environ.setdefault("MKL_NUM_THREADS", "1")

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


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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
                                 "xtol": 0.0, "updating": "deferred"})
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
                  f"host cpu_count={os.cpu_count()}, cpu model {cpu_model()}",
        "cpu_model": cpu_model(),
        "t_first_eval_s": gens[0] - t0,
    }


def minimize_wall(objective, n, P, strategy, maxiter=1000):
    """evals/s = nit * popsize / wall around the WHOLE minimize() call (host-side Latin
    hypercube, uploads, result download included) after one warm-up call; ftol=-1, xtol=0 so the run lasts maxiter."""
    import torch

    import stochopy_amd as sa

    opts = {"maxiter": maxiter, "popsize": P, "seed": 0, "strategy": strategy, "ftol": -1.0, "xtol": 0.0,
            "updating": "deferred", "rng": "philox", "backend": "hip"}
    fun = getattr(sa.factory, objective)
    bounds = [[-5.12, 5.12]] * n
    sa.optimize.minimize(fun, bounds, method="de", options=dict(opts, maxiter=50))
    walls = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sa.optimize.minimize(fun, bounds, method="de", options=opts)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
    wall = min(walls)
    return {"value": res.nit * P / wall, "unit": "evals/s", "wall_s": wall, "walls_s": walls, "nit": int(res.nit),
            "nfev": int(res.nfev),
            "note": "wall clock around stochopy_amd.optimize.minimize(method='de', updating='deferred', rng='philox'), "
                    "best of three calls after one warm-up call; everything included: the initial population (drawn on the "
                    "device since round 3), uploads, graph replays, the result copy"}


def eval_kernel_config(objective, n, P, reps=50):
    """sx_eval alone: HIP events on the engine stream around `reps` launches over a resident (P, n) array; (8n + 8) B per evaluation."""
    import torch

    from stochopy_amd import _device, _lib

    ctx = _device.Context()
    X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
    f = ctx.empty((P,))
    fid = _lib.FUN_IDS[objective]
    torch.cuda.synchronize()
    with torch.cuda.stream(ctx.stream):
        for _ in range(5):
            _device.evaluate(ctx, fid, X, n, f=f)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        for _ in range(reps):
            _device.evaluate(ctx, fid, X, n, f=f)
        e1.record(ctx.stream)
        ctx.sync()
    us = e0.elapsed_time(e1) / reps * 1e3
    return {"evals_per_s": P / us * 1e6, "kernel_us": us, "bound": "hbm", "frac": (8 * n + 8) * P / us / 1e3 / HBM_PEAK_GBS,
            "note": "the objective kernel alone (sx_eval), HIP events around %d launches; (8 n + 8) B per evaluation" % reps}


def other_configs():
    """BASELINE.json configs 2-4 from the same process: evals/s = popsize / (wall per generation),
    wall per generation from TWO whole minimize() calls of different length (set-up and result copy cancel), best of
    three; `frac` = algorithmic bytes (or flops) of one generation / that time / peak -- a whole-generation figure, all
    kernels and gaps included, not a kernel figure."""
    import torch

    import stochopy_amd as sa

    def per_gen(method, fun, n, opts, short, long_, reps=3):
        bounds = [[-5.12, 5.12]] * n
        o = dict(dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip"), **opts)

        def wall(m):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = sa.optimize.minimize(fun, bounds, method=method, options=dict(o, maxiter=m))
            torch.cuda.synchronize()
            return time.perf_counter() - t0, r.nit

        # two run lengths, the fastest run of each (a disturbance only ever ADDS time: the minimum of the per-repetition
        # DIFFERENCES would reward a slow short run), and their difference per generation
        wall(short)
        runs = [(wall(short), wall(long_)) for _ in range(reps)]
        (t1, n1), (t2, n2) = min(r[0] for r in runs), min(r[1] for r in runs)
        return (t2 - t1) / (n2 - n1)

    out = {}
    mfma_f64 = 78.6e12  # dense fp64 MFMA peak used by SURVEY.md section 8d
    t = per_gen("de", sa.factory.rastrigin, 128, {"popsize": 4096, "updating": "deferred", "strategy": "best1bin"}, 200, 2200)
    out["C2_de_rastrigin_n128_p4096"] = {"evals_per_s": 4096 / t, "us_per_generation": t * 1e6, "bound": "hbm",
                                        "frac": 4112 * 4096 / t / (HBM_PEAK_GBS * 1e9)}
    # BASELINE config 5 on ONE GPU (round 4): its 8-GPU shard (16 384 rows: the denominator-free per-GPU figure of a weak-
    # scaling read) and the whole population (131 072 rows = 2 x 1 GiB of row buffers: the N=1 point of the strong-
    # scaling curve the driver's N = 1, 2, 4, 8 runs draw through `c5`)
    de5 = {"updating": "deferred", "strategy": "best1bin"}
    t = per_gen("de", sa.factory.rosenbrock, 1024, dict(de5, popsize=16384), 100, 600)
    out["C5_shard_de_n1024_p16384"] = {"evals_per_s": 16384 / t, "us_per_generation": t * 1e6, "bound": "hbm",
                                      "frac": 32784 * 16384 / t / (HBM_PEAK_GBS * 1e9)}
    t = per_gen("de", sa.factory.rosenbrock, 1024, dict(de5, popsize=131072), 20, 120, reps=2)
    out["C5_full_de_n1024_p131072_1gpu"] = {"evals_per_s": 131072 / t, "us_per_generation": t * 1e6, "bound": "hbm",
                                           "frac": 32784 * 131072 / t / (HBM_PEAK_GBS * 1e9)}
    # the metric shape with the reference's own random stream (rng="numpy-legacy": seed-for-seed the reference's run,
    # tests/golden/configs.json M_de_rosen_n128_p4096): host-bound by the replay of numpy's MT19937 donor permutations
    t = per_gen("de", sa.factory.rosenbrock, 128, {"popsize": 4096, "updating": "deferred", "strategy": "best1bin",
                                                   "rng": "numpy-legacy"}, 5, 25, reps=2)
    out["M_numpy_legacy_de_rosenbrock_n128_p4096"] = {
        "evals_per_s": 4096 / t, "ms_per_generation": t * 1e3, "bound": "host",
        "note": "parity mode: same seed => the reference's trajectory bit for bit; per generation the host replays 23 M words of "
                "numpy's legacy MT19937 stream (de/_de.py:304-311, P permutations of P-1 indices) -- csrc/sx_mt19937.cpp"}
    # the metric's row length WITHOUT the launch-bound effects: DE at n = 128 with 2^20 individuals
    # (4.3 GB algorithmic per generation, 1 GiB per population buffer: nothing of it fits a cache), and the objective
    # kernel alone (rows a1 / a5 of SURVEY.md section 8a: the literal "objective-fn evals/s") at the same shape
    t = per_gen("de", sa.factory.rosenbrock, 128, dict(de5, popsize=1 << 20), 10, 60, reps=2)
    out["M_large_de_rosenbrock_n128_p1048576"] = {"evals_per_s": (1 << 20) / t, "us_per_generation": t * 1e6, "bound": "hbm",
                                                  "frac": 4112 * (1 << 20) / t / (HBM_PEAK_GBS * 1e9)}
    out["eval_rosenbrock_n128_p1048576"] = eval_kernel_config("rosenbrock", 128, 1 << 20)
    out["eval_ackley_n256_p524288"] = eval_kernel_config("ackley", 256, 1 << 19)
    # rows beyond 4096 elements (round 5: csrc/sx_wide.hip) where VD-CMA is the method of choice
    t = per_gen("vdcma", sa.factory.rosenbrock, 16384, {"popsize": 1024, "sigma": 0.3}, 10, 60, reps=2)
    out["VD_vdcma_rosenbrock_n16384_p1024"] = {
        "evals_per_s": 1024 / t, "us_per_generation": t * 1e6, "bound": "hbm", "frac": 32 * 16384 * 1024 / t / (HBM_PEAK_GBS * 1e9),
        "note": "algorithmic bytes 32 n per candidate: y and x written (16 n), x read by the objective (8 n), x and y of the "
                "mu = P/2 selected rows read by the moment sums (8 n per candidate on average); DESIGN.md section 4.  Since "
                "round 6 the device-resident run does not keep x (the moment sums form it again from y): the kernels move "
                "~24 n per candidate incl. the normals parked between a streamed row's two passes; the basis of `frac` stays "
                "the 32 n of rounds 5's line"}
    c3 = {"popsize": 16384, "updating": "deferred"}
    t = per_gen("pso", sa.factory.ackley, 256, c3, 200, 1200)
    out["C3a_pso_ackley_n256_p16384"] = {"evals_per_s": 16384 / t, "us_per_generation": t * 1e6, "bound": "hbm",
                                        "frac": 12312 * 16384 / t / (HBM_PEAK_GBS * 1e9)}
    t = per_gen("cpso", sa.factory.ackley, 256, c3, 200, 1200)
    out["C3b_cpso_ackley_n256_p16384"] = {"evals_per_s": 16384 / t, "us_per_generation": t * 1e6, "bound": "hbm",
                                         "frac": 12312 * 16384 / t / (HBM_PEAK_GBS * 1e9),
                                         "note": "PSO's bytes per evaluation; the competitive restart's own traffic is extra"}
    t = per_gen("cmaes", sa.factory.rosenbrock, 512, {"popsize": 1024}, 10, 50, reps=2)
    flops = 2.0 * 1024 * 512**2 + 2.0 * 512 * 512**2
    out["C4_cmaes_rosenbrock_n512_p1024"] = {
        "evals_per_s": 1024 / t, "ms_per_generation": t * 1e3, "bound": "mfma", "frac": flops / t / mfma_f64,
        "note": "generations 10-50 of a run (every generation decomposes the covariance: 5-6 Jacobi sweeps, then -- round 6 -- one or "
                "two deep refinement steps (an exact similarity with exp(K) on the matrix cores) and the first-order one); frac = sampling + rank-mu flops (8.05e8) / generation time / 78.6 TF -- the generation "
                "is the eigendecomposition's latency chain, not these two contractions.  Parity of this mode (device eigensolver, Philox "
                "draws): every single generation is pinned to the oracle's model of that generation and runs agree with the oracle "
                "(LAPACK + canonical signs) within 1e-6 over windows of 16-24 generations; whole-run agreement at this size is a statement "
                "about a seed-dependent number of generations for ANY two eigensolvers (profiles/r4_c4_parity_margin.txt)"}
    return out


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

    cores = os.cpu_count() or 1  # workers = os.cpu_count(), as SURVEY.md section 8d asks (pool warm before timing)
    stamps = []
    with Parallel(n_jobs=cores, backend="loky") as pool:
        def fobj(X):
            return np.array(pool(delayed(_one_row)(objective, x) for x in X))

        t_pool = time.perf_counter()
        pool(delayed(_one_row)(objective, np.zeros(n)) for _ in range(4 * cores))  # start every worker
        t_pool = time.perf_counter() - t_pool

        def cb(X, r):
            stamps.append(time.perf_counter())
            if len(stamps) >= 3 and stamps[-1] - stamps[0] > budget_s:
                raise StopIteration

        try:
            oracle.minimize(fobj, [[-5.12, 5.12]] * n, method="de", callback=cb,
                            options={"maxiter": 10**6, "popsize": P, "seed": 0, "strategy": strategy, "ftol": -1.0,
                                     "xtol": 0.0, "updating": "deferred"})
        except StopIteration:
            pass
    done, dt = len(stamps) - 1, stamps[-1] - stamps[0]
    return {"value": P * done / dt, "unit": "evals/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"oracle DE {strategy} {objective} n={n} P={P}, {done} generations in {dt:.1f}s, one joblib-loky task "
                      f"per individual on {cores} workers = os.cpu_count() (the reference's parallel backend scheme; "
                      f"pool started and warmed in {t_pool:.1f}s before timing), cpu model {cpu_model()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="de_rosenbrock_n128_p4096", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-minimize-wall", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the C2 / C3a / C3b / C4 block")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=12.0, help="CPU work spent on the baseline sample")
    ap.add_argument("--kernel-timing-launches", type=int, default=400)
    ap.add_argument("--min-timed-seconds", type=float, default=2.0,
                    help="the K-step block is repeated until the timed region lasts this long (profilers: pass less)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: become `torch.distributed.run --nproc-per-node N bench.py ...`
        # (one process per GPU, rendezvous on 127.0.0.1: the container hostname may not resolve)
        import socket

        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        os.execv(sys.executable, [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                                  f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
                                  os.path.abspath(__file__)] + sys.argv[1:])

    import torch

    import stochopy_amd as sa  # noqa: F401  (fails loudly here if the HIP library is missing)
    from stochopy_amd import _lib
    from stochopy_amd.optimize import _de

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one process per GPU "
                         "(or call bench.py without a launcher: it starts torch.distributed.run itself)")
    # (test switches, one-GPU boxes only: SX_BENCH_DEVICE pins every rank to one device, SX_BENCH_BACKEND=gloo sets the
    #  process group up without RCCL, which refuses two ranks on one GPU; the driver uses neither)
    device_index = int(os.environ.get("SX_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(device_index)
    dist = None
    if world > 1 or ("RANK" in os.environ and os.environ.get("SX_FORCE_SHARDED") == "1"):
        import torch.distributed as dist

        # torch's NCCL flight recorder on: graph captures of the RCCL transport wait for the watchdog to RETIRE earlier
        # collectives by reading it (parallel.World.quiesce_for_capture) instead of sleeping
        os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "256")  # (TORCH_NCCL_TRACE_BUFFER_SIZE before torch 2.8)
        os.environ.setdefault("TORCH_NCCL_TRACE_BUFFER_SIZE", "256")
        backend = os.environ.get("SX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
        else:
            dist.init_process_group(backend)

    objective, n, P, strategy = WORKLOADS[args.workload]
    k = _lib.DE_DONORS[strategy]
    K, W = args.steps, args.warmup
    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(x, dtype):
        """MAX over ranks of one host number (on the device for nccl, on the host for gloo)."""
        dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
        tt = torch.tensor([x], dtype=dtype, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return tt.item()

    MAX_BLOCKS = 1 << 20
    MIN_TIMED_S = args.min_timed_seconds  # (the driver's GPU-busy sampler and timed-region check need seconds)

    def self_check(exchange, gens=3):
        """First contact with distinct devices validates itself before anything is timed (VERDICT r5 next #6c): `gens`
        generations of the benchmark's own run over `exchange`, then every rank's (best f, global row, generation count) --
        they must be ONE answer on all ranks.  Returns it (the caller compares the transports with each other)."""
        lower, upper = np.full(n, -5.12), np.full(n, 5.12)
        run = _de._DeRun(_lib.FUN_IDS[objective], lower, upper, None, 2**31 - 2, world * P, 0.5, 0.9, strategy,
                         None, 0.0, -1.0, False, 1.0, None, "philox", 1234, world, autorun=False, exchange=exchange,
                         donors=os.environ.get("SX_DONORS") or "shard")
        try:
            with torch.cuda.stream(run.ctx.stream):
                run._setup()
                run.enqueue(gens)
                st = run.read_state()
            mine = (float(st.gfit).hex(), int(st.gbidx), int(st.it))
            everyone = [None] * world
            dist.all_gather_object(everyone, mine)
            barrier()  # no rank frees its exchange buffer while a peer may still write into it
        finally:
            run.close()
        if any(e != everyone[0] for e in everyone):
            raise RuntimeError(f"self-check over exchange={exchange!r}: the ranks disagree after {gens} generations: {everyone}")
        return everyone[0]

    def measure(exchange, objective=objective, n=n, Ptotal=None, strategy=strategy, donors=os.environ.get("SX_DONORS") or "shard",
                K=K, W=W, kernel_launches=args.kernel_timing_launches):
        """One resident run: W warm-up steps, then blocks of K steps until >= 2 s are timed.  Ptotal None: weak
        scaling (every GPU owns P rows of a global population of world*P; same seed on every rank)."""
        lower, upper = np.full(n, -5.12), np.full(n, 5.12)
        run = _de._DeRun(_lib.FUN_IDS[objective], lower, upper, None, 2**31 - 2, Ptotal or world * P, 0.5, 0.9, strategy,
                         None, 0.0, -1.0, False, 1.0, None, "philox", 1234, world, autorun=False, exchange=exchange,
                         donors=donors)
        ctx = run.ctx
        try:
            with torch.cuda.stream(ctx.stream):
                run._setup()
                run.prepare_graphs()
                run.enqueue(W)
                # two more untimed K-step blocks: a block length that repeats gets a graph of its own on its second request
                # (the chained kernel's tails; the RCCL transport's kernels + all-gathers, whose capture first waits for the
                # NCCL watchdog to go idle: ~1 s) -- that happens here, not inside the timed region
                run.enqueue(K)
                run.enqueue(K)
                primed = 2 * K
                ctx.sync()
                # K-step blocks, back to back, until the timed region lasts >= 2 s (every rank times the same number)
                blocks, steps_done = 1, 0
                while True:
                    # an event record costs the stream ~2.6 us: one every >= 200 steps (every block when K >= 200)
                    group = max(1, 200 // K)
                    evs = [torch.cuda.Event(enable_timing=True) for _ in range(blocks // group + 1)]
                    barrier()
                    t0 = time.perf_counter()
                    evs[0].record(ctx.stream)
                    for b in range(blocks):
                        run.enqueue(K)
                        if (b + 1) % group == 0:
                            evs[(b + 1) // group].record(ctx.stream)
                    ctx.sync()
                    barrier()
                    t1 = time.perf_counter()
                    steps_done += blocks * K
                    enough = t1 - t0 >= MIN_TIMED_S or blocks >= MAX_BLOCKS
                    if dist is not None:
                        enough = reduce_max(0 if enough else 1, torch.int64) == 0  # all ranks or none
                    if enough:
                        break
                    grow = int(np.ceil(1.3 * blocks * MIN_TIMED_S / max(t1 - t0, 1e-6)))
                    blocks = max(2 * blocks, grow)
                    if dist is not None:
                        blocks = int(reduce_max(blocks, torch.int64))
                    blocks = min(blocks, MAX_BLOCKS)
                st = run.read_state()  # raises if a wait inside the peer exchange timed out
                assert st.it == 1 + W + primed + steps_done, (st.it, W, K, blocks, steps_done)
                block_ms = sorted(evs[g].elapsed_time(evs[g + 1]) / group for g in range(blocks // group))
                block_ms = block_ms or [(t1 - t0) * 1e3 / blocks]  # fewer blocks than one event group

                # dominant kernel: HIP events on the engine stream around a replayed hipGraph of generation
                # kernels (real generations, nothing else on the stream), average per launch
                nl = max(1, kernel_launches // run.GRAPH_CHUNK) * run.GRAPH_CHUNK
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
        dt = t1 - t0
        if dist is not None:
            dt = float(reduce_max(dt, torch.float64))
        return {"run": run, "dt": dt, "steps_timed": blocks * K, "blocks": blocks,
                "block_ms_median": block_ms[len(block_ms) // 2], "kern_ms": kern_ms, "nl": nl,
                "kernels_per_gen": kernels_per_gen, "rows_total": Ptotal or world * P}

    # ---- N > 1: what the first contact with a second physical GPU must not cost ----------------
    # The RCCL transport (one all-gather per generation: the library's own, exercised path) is measured FIRST and its
    # line is ready before the peer-write transport (kernels of different processes writing into each other's HBM over
    # IPC mappings: never run across a physical link before the driver's SCALE run) is even negotiated.  From then on a
    # watchdog holds the line: if a later stage wedges -- a hang no timeout inside the exchange catches -- rank 0 still
    # prints ONE line (the RCCL measurement, with what wedged in `transport_fallback`) and every rank leaves.
    state = {"line": None, "stage": None, "deadline": None}

    def watchdog():
        while True:
            time.sleep(1.0)
            dl = state["deadline"]
            if dl is not None and time.monotonic() > dl:
                if rank == 0 and state["line"] is not None:
                    ln = dict(state["line"])
                    ln["transport_fallback"] = (f"stage '{state['stage']}' did not finish within its allowance: the line "
                                                f"is the RCCL measurement taken before it")
                    print(json.dumps(ln), flush=True)
                sys.stdout.flush()
                sys.stderr.flush()
                os._exit(0 if rank == 0 and state["line"] is not None else 3)

    def guarded(stage, seconds, fn):
        """Run fn() as `stage`; if it does not return within `seconds` the watchdog prints the line held so far."""
        state["stage"], state["deadline"] = stage, time.monotonic() + seconds
        try:
            return fn()
        finally:
            state["deadline"] = None

    def measure_p2p_or_none(stage, seconds, **kw):
        """The peer-write transport, or None with the reason (negotiation failed on some rank, or a wait timed out)."""
        try:
            m = guarded(stage, seconds, lambda: measure("p2p", **kw))
            return m, None
        except Exception as e:  # noqa: BLE001  (every rank takes the same way: negotiate() agrees, time-outs reach all)
            print(f"[bench] rank {rank}: {stage}: {e}", file=sys.stderr, flush=True)
            return None, str(e)[:300]

    transports = None
    checks = {}
    if world == 1:
        m = measure(None)
    else:
        import threading

        threading.Thread(target=watchdog, daemon=True).start()
        checks = {"rccl": guarded("self-check rccl", 300.0, lambda: self_check("rccl"))}  # (raises: nothing valid to time)
        m_rccl = guarded("rccl", 600.0, lambda: measure("rccl"))
        transports = {"rccl": {"value": m_rccl["rows_total"] * m_rccl["steps_timed"] / m_rccl["dt"],
                               "ms_per_step": m_rccl["dt"] / m_rccl["steps_timed"] * 1e3}}
        m = m_rccl
    def build_line(m, c5=None, transport_fallback=None):
        run, dt, kern_ms, nl, kernels_per_gen = m["run"], m["dt"], m["kern_ms"], m["nl"], m["kernels_per_gen"]
        value = m["rows_total"] * m["steps_timed"] / dt
        bytes_per_launch = algorithmic_bytes_per_eval(n, k) * P
        achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
        if os.path.exists(pmc):
            try:  # HBM bytes per launch from separate rocprofv3 --pmc passes of this command (tools/pmc.sh), with the
                rec = json.load(open(pmc))  # commit they were measured at: not measured inside this run
                traffic = rec.get(args.workload, {}).get("hbm_bytes_per_launch")
                traffic_src = "profiles/pmc_latest.json: rocprofv3 --pmc passes (tools/r6_profiles.sh) at commit %s" % rec.get(
                    "_commit", "unrecorded")
            except Exception:
                traffic = None
        line = {
            "metric": "objective-fn evals/sec",
            "value": value,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": dt / m["steps_timed"] * 1e3,
            "timed": {"blocks": m["blocks"], "steps_timed": m["steps_timed"], "seconds": dt,
                      "block_ms_median": m["block_ms_median"],
                      "note": "W warm-up steps and two untimed K-step blocks (the K-step graph is instantiated there), then the K-step block "
                              "repeated back to back until >= 2 s are timed (one barrier + synchronize pair around the region); block_ms_median from HIP events between blocks (one event per "
                              "max(1, 200 // K) blocks, divided by that count)"},
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
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "kernel_us": kern_ms * 1e3,
                "timing": "HIP events on the engine stream around %d generations (%d kernel(s) each)" % (nl, kernels_per_gen),
            },
        }
        # what actually ran, at the top level (the driver's SCALE run reads this, not the nested blocks): ranks seen, the
        # transport behind `value`, BOTH transports' values when N > 1, and why if the peer transport is not the one
        line["n_ranks_seen"] = dist.get_world_size() if dist is not None else 1
        line["process_group"] = dist.get_backend() if dist is not None else None
        line["transport"] = None if run.world is None else run.exchange
        line["transports"] = transports
        if run.world is not None:  # (3 generations per transport before anything was timed: one answer on every rank)
            line["self_check"] = {k: {"best_f": float.fromhex(v[0]), "global_row": v[1], "generation": v[2]} for k, v in checks.items()}
        line["transport_fallback"] = transport_fallback or getattr(run, "exchange_note", None)
        if c5 is not None:
            line["c5"] = c5
        return line

    # BASELINE config 5 with N > 1: DE n=1024, P=131072 IN TOTAL (strong scaling; the N = 1 point is
    # configs.C5_full_de_n1024_p131072_1gpu of the N = 1 line), shard-local donors through both transports and global
    # donors (which need the peer mappings) -- RCCL first here too, every stage under the watchdog
    c5 = None
    p2p_note = None
    if world > 1:
        c5kw = dict(objective="rosenbrock", n=1024, Ptotal=131072, strategy="best1bin", K=20, W=10, kernel_launches=50)
        c5 = {"workload": "de_rosenbrock_n1024_p131072 (total), best1bin", "n_ranks_seen": dist.get_world_size(),
              "rows_per_gpu": 131072 // world, "scaling": "strong",
              "n1_reference": "configs.C5_full_de_n1024_p131072_1gpu of the N = 1 line (profiles/r6_bench_N1.json)"}

        def c5_entry(r, mode):
            return {"value": 131072 * r["steps_timed"] / r["dt"], "unit": "evals/s",
                    "ms_per_step": r["dt"] / r["steps_timed"] * 1e3, "steps_timed": r["steps_timed"],
                    "exchange": r["run"].exchange, "exchange_note": getattr(r["run"], "exchange_note", None),
                    "semantics": ("island model with a shared global best (documented deviation)" if mode == "shard" else
                                  "the unsharded run: donor rows read from their owners' HBM over xGMI")}

        if rank == 0:
            state["line"] = build_line(m_rccl)
        try:
            r = guarded("c5 shard-local donors over rccl", 600.0, lambda: measure("rccl", donors="shard", **c5kw))
            c5["donors_shard_rccl"] = c5_entry(r, "shard")
        except Exception as e:  # noqa: BLE001
            c5["donors_shard_rccl"] = {"error": str(e)[:300]}
        if rank == 0:
            state["line"] = build_line(m_rccl, c5)
        p2p_off = ("SX_BENCH_P2P=0" if os.environ.get("SX_BENCH_P2P", "1") == "0" else
                   "SX_EXCHANGE=rccl" if os.environ.get("SX_EXCHANGE") == "rccl" else None)
        if p2p_off is None:
            # the peer-write transport must give the RCCL transport's answer before it is timed: same (f, row, generation)
            try:
                checks["p2p"] = guarded("self-check p2p", 300.0, lambda: self_check("p2p"))
                if checks["p2p"] != checks["rccl"]:
                    p2p_off = f"self-check: p2p gave {checks['p2p']}, rccl {checks['rccl']} after 3 generations"
            except Exception as e:  # noqa: BLE001  (negotiation failed on some rank, a wait timed out, or the ranks disagree)
                p2p_off = "self-check p2p: " + str(e)[:300]
        if p2p_off is None:
            m_p2p, p2p_note = measure_p2p_or_none("p2p (peer writes over xGMI)", 300.0)
            if m_p2p is not None:
                v = m_p2p["rows_total"] * m_p2p["steps_timed"] / m_p2p["dt"]
                transports["p2p"] = {"value": v, "ms_per_step": m_p2p["dt"] / m_p2p["steps_timed"] * 1e3}
                if v > transports["rccl"]["value"]:
                    m = m_p2p
                if rank == 0:
                    state["line"] = build_line(m, c5)
                for mode in ("shard", "global"):
                    r, note = measure_p2p_or_none(f"c5 {mode} donors over p2p", 300.0, donors=mode, **c5kw)
                    c5[f"donors_{mode}_p2p"] = c5_entry(r, mode) if r is not None else {"error": note}
                    if rank == 0:
                        state["line"] = build_line(m, c5)
            else:
                transports["p2p"] = {"error": p2p_note}
                c5["donors_global_p2p"] = {"error": "needs the peer mappings: " + (p2p_note or "")}
        else:
            p2p_note = p2p_off + ("" if p2p_off.startswith("self-check") else " (switched off by the environment)")
            transports["p2p"] = {"skipped": p2p_note}
            c5["donors_global_p2p"] = {"error": "needs the peer exchange: " + p2p_note}
        state["deadline"] = None

    run = m["run"]
    value = m["rows_total"] * m["steps_timed"] / m["dt"]
    if rank == 0:
        line = build_line(m, c5, None if (transports is None or "value" in transports.get("p2p", {})) else
                          f"peer-write transport not used: {p2p_note}")
        if world == 1 and not args.no_minimize_wall:
            line["minimize_wall"] = minimize_wall(objective, n, P, strategy)
        if world == 1 and not args.no_configs:
            try:
                line["configs"] = other_configs()
            except Exception as e:  # noqa: BLE001  (keep the headline line whatever happens here)
                line["configs"] = {"error": str(e)[:300]}
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
