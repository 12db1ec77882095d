#!/usr/bin/env python3
"""Times the hand-written block-Jacobi eigensolver (sx_eigh) against torch.linalg.eigh (rocSOLVER syevd) on
CMA-like covariances; prints sweeps, residual and orthogonality.  usage: bench_eigh_jacobi.py [n ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stochopy_amd import _device  # noqa: E402
from stochopy_amd.linalg import Eigh  # noqa: E402

sizes = [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024]
ctx = _device.Context()


def cma_like(n, gens, rs):
    mu = n
    w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    w /= w.sum()
    mueff = 1 / np.sum(w**2)
    c1 = 2.0 / ((n + 1.3) ** 2 + mueff)
    cmu = min(1.0 - c1, 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))
    C = np.eye(n)
    for _ in range(gens):
        d, B = np.linalg.eigh(C)
        Y = (rs.randn(mu, n) * np.sqrt(d)) @ B.T
        C = (1 - c1 - cmu) * C + cmu * (Y.T * w) @ Y
        C = np.triu(C) + np.triu(C, 1).T
    return C


def timed(fn, reps):
    with torch.cuda.stream(ctx.stream):
        fn()
        ctx.sync()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        for _ in range(reps):
            fn()
        e1.record(ctx.stream)
        ctx.sync()
    return e0.elapsed_time(e1) / reps


for n in sizes:
    rs = np.random.RandomState(n)
    for kind in ("cma", "spd"):
        if kind == "cma":
            Cm = cma_like(n, 3, rs)
        else:
            A = rs.randn(n, n)
            Cm = A @ A.T / n + 0.1 * np.eye(n)
        dC = ctx.upload(Cm)
        eig = Eigh(ctx, n)
        eig(dC)
        sweeps, conv, off = eig.info()
        ms = timed(lambda: eig(dC, max_sweeps=sweeps + 1), 5)
        w, B = eig(dC)
        ctx.sync()
        w, B = w.cpu().numpy(), B.cpu().numpy()
        nC = np.linalg.norm(Cm)
        resid = np.linalg.norm(Cm - (B * w) @ B.T) / nC
        orth = np.abs(B.T @ B - np.eye(n)).max()
        wr = np.linalg.eigvalsh(Cm)
        with torch.cuda.stream(ctx.stream):
            ms_t = timed(lambda: torch.linalg.eigh(dC), 3)
        print(f"n={n:5d} {kind}: sx_eigh {ms:8.3f} ms ({sweeps} sweeps, conv={conv}, off={off:.1e}) "
              f"resid {resid:.1e} orth {orth:.1e} eig {np.abs(w - wr).max() / np.abs(wr).max():.1e} | "
              f"torch.linalg.eigh {ms_t:8.3f} ms", flush=True)
