#!/usr/bin/env python3
"""Randomised parity of the round-2 paths against the oracle / LAPACK (GPU box):
  eigh   random symmetric matrices (sizes 1..600, several spectra, cold and warm) vs numpy.linalg.eigh
  cmaes  the device-resident loop vs the oracle with LAPACK + canonical signs (mu + 1 >= n shapes)
  vdcma  the device-resident VD-CMA loop vs the oracle
  na     the Neighbourhood Algorithm vs the oracle, both rng modes (bit for bit on + - * objectives)
  host   plain Python objectives vs the fused run (bit for bit when the objective has the kernel's bits)
usage: fuzz_round2.py [seconds per family]"""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import oracle
from oracle import engine as oe
import stochopy_amd as sa
from stochopy_amd import _device
from stochopy_amd.linalg import Eigh

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rs = np.random.RandomState(int(os.environ.get("FUZZ_SEED", "20260928")))
ctx = _device.Context()
warnings.simplefilter("ignore", RuntimeWarning)


def family(name, one):
    t0, runs, bad = time.time(), 0, 0
    while time.time() - t0 < budget:
        ok, what = one()
        runs += 1
        if not ok:
            bad += 1
            print(f"  MISMATCH {name}: {what}", flush=True)
    print(f"{name}: {runs} runs, {bad} mismatches", flush=True)
    return bad


def one_eigh():
    n = int(rs.choice([rs.randint(1, 65), rs.randint(65, 200), rs.randint(200, 600)], p=[0.4, 0.4, 0.2]))
    kind = rs.choice(["spd", "indef", "graded", "cluster"])
    A = rs.randn(n, n)
    if kind == "spd":
        C = A @ A.T / n + rs.uniform(0, 1) * np.eye(n)
    elif kind == "indef":
        C = A + A.T
    else:
        Q = np.linalg.qr(A)[0]
        lam = np.logspace(0, -rs.uniform(1, 10), n) if kind == "graded" else 1.0 + 1e-3 * rs.randn(n)
        C = (Q * lam) @ Q.T
    C *= 10.0 ** rs.uniform(-6, 6)
    Cs = np.triu(C) + np.triu(C, 1).T
    with torch.cuda.stream(ctx.stream):
        eig = Eigh(ctx, n)
        start = None
        if n > 64 and rs.rand() < 0.5:  # warm start from the eigenvectors of a nearby matrix
            E = rs.randn(n, n) * 1e-3 * np.abs(Cs).max() / np.sqrt(n)
            start = ctx.upload(np.linalg.eigh(Cs + 0.5 * (E + E.T))[1])
        w, B = eig(ctx.upload(C), max_sweeps=48, start=start)
        w, B = w.cpu().numpy(), B.cpu().numpy()
        sweeps, conv, _ = eig.info()
    wr = np.linalg.eigvalsh(Cs)
    nC = max(np.linalg.norm(Cs), 1e-300)
    e1 = np.abs(w - wr).max() / max(np.abs(wr).max(), 1e-300)
    e2 = np.linalg.norm(Cs / nC - (B * (w / nC)) @ B.T)
    e3 = np.abs(B.T @ B - np.eye(n)).max()
    # Limits.  Full sweeps: residual 1e-13 sqrt(n)/4 |C|, orthogonality 1e-13.  With the refinement step allowed
    # (SX_EIGH_REFINE=1: what the CMA-ES loops run with) the rule that admits the step bounds what it leaves behind by
    # ~0.5 max|K| off <= 5e-13 |C| (csrc/sx_eigh.hip kRefineProd): residual 2e-12 |C| (the limit
    # tests/test_gpu_eigh.py::test_eigh_with_the_refinement_step states); orthogonality 1e-12: the step's two products on a
    # graded spectrum (condition up to 1e10, 15+ sweeps before it) were seen at 3e-13 ... 5e-13 twice in 3 211 runs
    # (profiles/r4_c4_parity_margin.txt), everything else below 2e-13.
    refine = os.environ.get("SX_EIGH_REFINE") == "1"
    res_lim = 2e-12 if refine else 1e-13 * max(1.0, np.sqrt(n) / 4)
    ok = conv and e1 <= (2e-12 if refine else 1e-12) and e2 <= res_lim and e3 <= (1e-12 if refine else 1e-13)
    return ok, f"n={n} {kind} warm={start is not None} sweeps={sweeps} conv={conv} eig={e1:.1e} resid={e2:.1e} orth={e3:.1e}"


tie_cases = 0


def one_cmaes():
    big = rs.rand() < 0.15  # round 3: now and then the block path of the eigensolver (n > 64) inside the loop
    n = int(rs.randint(65, 160)) if big else int(rs.randint(2, 40))
    P = int(rs.randint(2 * n + 2, 4 * n + 8))  # mu + 1 >= n: the eigenbasis is determined
    obj = str(rs.choice(["rosenbrock", "sphere", "rastrigin", "ackley"]))
    o = {"maxiter": int(rs.randint(3, 14 if big else 40)), "popsize": P, "seed": int(rs.randint(1 << 30)),
         "sigma": float(rs.uniform(0.05, 0.5))}
    if rs.rand() < 0.3:
        o["return_all"] = True
        o["verbosity"] = float(rs.choice([0.0, 0.5, 1.0]))
    b = [[-float(rs.uniform(1, 6)), float(rs.uniform(1, 6))]] * n
    if rs.rand() < 0.35:  # round 3: constraints="Penalize" stays in the device-resident loop; boxes the mean tends to leave
        o["constraints"] = "Penalize"
        lo = float(rs.uniform(-2, 2))
        b = [[lo, lo + float(rs.uniform(0.5, 4))]] * n
    ref = oracle.minimize(obj, b, method="cmaes", options=dict(o, eigh="canonical"), rng="philox")
    if "constraints" in o:
        # Candidates clipped to the same corner of the box have EXACTLY equal fitness while the boundary weights are still
        # zero.  The device ranks ties by index (stable); numpy's default argsort -- the reference's and the oracle's -- is
        # the AVX-512 / AVX2 vectorised sort on x86, whose tie order is an accident of the build and the CPU.  Such runs have
        # no defined reference trajectory: counted, not compared.
        full = oracle.minimize(obj, b, method="cmaes", options=dict(o, eigh="canonical", return_all=True, verbosity=1.0), rng="philox")
        if any(len(np.unique(row)) < len(row) for row in full.funall):
            global tie_cases
            tie_cases += 1
            return True, "exact fitness ties (undefined order in the reference)"
    got = sa.optimize.minimize(getattr(sa.factory, obj), b, method="cmaes", options=dict(o, backend="hip", rng="philox"))
    ok = (got.nit, got.status) == (ref.nit, ref.status) and np.isclose(got.fun, ref.fun, rtol=1e-5, atol=1e-12)
    if ok and "return_all" in o:
        ok = got.funall.shape == ref.funall.shape and np.allclose(got.funall, ref.funall, rtol=1e-5, atol=1e-12)
    return ok, f"{obj} n={n} P={P} {o}: got {got.fun!r}/{got.nit}/{got.status} ref {ref.fun!r}/{ref.nit}/{ref.status}"


def one_vdcma():
    n = int(rs.choice([6, 7, 9, 16, 33, 64, 100, 257, 700]))
    P = int(rs.randint(4, 40))
    obj = str(rs.choice(["rosenbrock", "sphere", "rastrigin"]))
    o = {"maxiter": int(rs.randint(3, 80)), "popsize": P, "seed": int(rs.randint(1 << 30)), "sigma": float(rs.uniform(0.05, 0.5)),
         "muperc": float(rs.choice([0.25, 0.5, 1.0]))}
    if rs.rand() < 0.3:
        o["return_all"] = True
        o["verbosity"] = float(rs.choice([0.0, 0.5, 1.0]))
    if rs.rand() < 0.3:
        o["ftol"] = float(10 ** rs.uniform(-6, 2))
    b = [[-float(rs.uniform(1, 6)), float(rs.uniform(1, 6))]] * n
    ref = oracle.minimize(obj, b, method="vdcma", options=dict(o), rng="philox")
    got = sa.optimize.minimize(getattr(sa.factory, obj), b, method="vdcma", options=dict(o, backend="hip", rng="philox"))
    ok = (got.nit, got.status) == (ref.nit, ref.status) and np.isclose(got.fun, ref.fun, rtol=1e-5, atol=1e-12)
    if ok and "return_all" in o:
        ok = got.funall.shape == ref.funall.shape and np.allclose(got.funall, ref.funall, rtol=1e-5, atol=1e-12)
    return ok, f"{obj} n={n} P={P} {o}: got {got.fun!r}/{got.nit}/{got.status} ref {ref.fun!r}/{ref.nit}/{ref.status}"


def one_na():
    n = int(rs.randint(1, 9))
    P = int(rs.randint(2, 30))
    obj = str(rs.choice(["rosenbrock", "sphere"])) if n > 1 else "sphere"
    o = {"maxiter": int(rs.randint(2, 30)), "popsize": P, "seed": int(rs.randint(1 << 30)), "nrperc": float(rs.uniform(0.05, 1.0)),
         "return_all": True, "verbosity": float(rs.choice([0.0, 0.5, 1.0]))}
    if rs.rand() < 0.3:
        o.update(ftol=float(10 ** rs.uniform(-3, 1)))
    b = [[-float(rs.uniform(1, 6)), float(rs.uniform(1, 6))] for _ in range(n)]
    if n > 1 and rs.rand() < 0.2:
        b[int(rs.randint(n))] = [0.75, 0.75]  # a fixed axis
    mode = str(rs.choice(["philox", "numpy-legacy"]))
    ref = oracle.minimize(obj, b, method="na", options=dict(o), rng=mode)
    got = sa.optimize.minimize(getattr(sa.factory, obj), b, method="na", options=dict(o, backend="hip", rng=mode))
    ok = ((got.nit, got.status) == (ref.nit, ref.status) and got.fun == ref.fun and np.array_equal(got.x, ref.x)
          and np.array_equal(got.xall, ref.xall) and np.array_equal(got.funall, ref.funall))
    return ok, f"{obj} n={n} P={P} {mode} {o} bounds {b}"


def one_host():
    n = int(rs.randint(1, 20))
    P = int(rs.randint(6, 40))
    method = str(rs.choice(["de", "pso", "cpso"]))
    mode = str(rs.choice(["philox", "numpy-legacy"]))
    o = {"maxiter": int(rs.randint(2, 25)), "popsize": P, "seed": int(rs.randint(1 << 30)), "updating": "deferred", "rng": mode,
         "backend": "hip"}
    b = [[-5.12, 5.12]] * n
    mine = sa.optimize.minimize(lambda x, k: k * np.sum(x**2), b, args=(1.0,), method=method, options=dict(o))
    fused = sa.optimize.minimize(sa.factory.sphere, b, method=method, options=dict(o))
    ok = np.array_equal(mine.x, fused.x) and mine.fun == fused.fun and (mine.nit, mine.status) == (fused.nit, fused.status)
    return ok, f"{method} n={n} P={P} {mode}"


bad = 0
only = os.environ.get("FUZZ_ONLY")
for name, fn in (("eigh", one_eigh), ("cmaes device loop", one_cmaes), ("vdcma device loop", one_vdcma), ("na", one_na),
                 ("host callable", one_host)):
    if only and only not in name:
        continue
    bad += family(name, fn)
print(f"(Penalize runs with exact fitness ties -- undefined argsort order in the reference, not compared: {tie_cases})")
print("TOTAL mismatches:", bad)
sys.exit(1 if bad else 0)
