#!/usr/bin/env python3
"""Per-sweep off-diagonal mass of the block-Jacobi eigensolver on hard spectra (debug/tuning helper).
usage: eigh_history.py kind n [max_sweeps]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from stochopy_amd import _device
from stochopy_amd.linalg import Eigh
import test_gpu_eigh as T

ctx = _device.Context()
for spec in sys.argv[1:]:
    kind, n, ms = (spec.split(":") + ["40"])[:3]
    n, ms = int(n), int(ms)
    rs = np.random.RandomState(77 + n)
    Cm = T.make(kind, n, rs)
    eig = Eigh(ctx, n)
    w, B = eig(ctx.upload(Cm), max_sweeps=ms)
    sweeps, conv, off = eig.info()
    raw = eig.ws[:8 + 60].cpu().numpy()
    norm2 = raw[2]
    acc = raw[4:4 + 60]
    hist = " ".join("%.1e" % np.sqrt(a / norm2) for a in acc[:max(sweeps, 1) + 1])
    w, B = w.cpu().numpy(), B.cpu().numpy()
    Cs = np.triu(Cm) + np.triu(Cm, 1).T
    resid = np.linalg.norm(Cs - (B * w) @ B.T) / np.linalg.norm(Cs)
    print(f"{kind} n={n}: sweeps {sweeps} conv {conv} resid {resid:.1e} orth {np.abs(B.T @ B - np.eye(n)).max():.1e}\n   off/|C| per sweep: {hist}", flush=True)
