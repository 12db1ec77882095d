"""Sweeps per decomposition inside a device-resident CMA-ES run (VERDICT r2 next #1a): drives the loop one generation
at a time, reads the eigensolver's EighInfo after each, prints the per-sweep off-diagonal mass and the generation's
GPU time.  usage: eigh_c4_sweeps.py [n P gens [max_sweeps]]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import stochopy_amd as sa
from stochopy_amd.optimize import _cmaes

n, P, gens = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 1024, 40)
ms = int(sys.argv[4]) if len(sys.argv) > 4 else 24
lo, up = np.full(n, -5.12), np.full(n, 5.12)
run = _cmaes._CmaDeviceRun(sa.factory.rosenbrock.sx_id, lo, up, None, gens + 1, P, 0.1, 0.5, 0.0, -1.0, 0, run=False)
run.args.eig_sweeps = ms
eigeneval, tot = 0, []
for gen in range(1, gens + 1):
    due = gen * P - eigeneval > run.eig_every
    if due:
        due = 2 if eigeneval else 1
        eigeneval = gen * P
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(run.ctx.stream):
        e0.record(run.ctx.stream)
        run.step(gen, int(due))
        e1.record(run.ctx.stream)
    run.ctx.sync()
    dt = e0.elapsed_time(e1)
    if due:
        raw = run.eig.ws[:128].cpu().numpy()
        hdr = raw[:2].view(np.int32)
        sweeps, conv = int(hdr[1]), int(hdr[3])
        norm2 = raw[2]
        acc = raw[4:4 + 60]
        offm = raw[64:64 + 60]
        hist = " ".join("%.1e" % np.sqrt(a / norm2) for a in acc[:max(sweeps, 1) + 1])
        left = " ".join("%.1e" % np.sqrt(a / norm2) for a in offm[:max(sweeps, 1)])
        print(f"gen {gen:3d} due {int(due)} {dt:7.3f} ms  sweeps {sweeps:2d} conv {conv}  off/|C| met per sweep: {hist} | left behind: {left}", flush=True)
        tot.append((dt, sweeps))
    else:
        print(f"gen {gen:3d} no decomposition {dt:7.3f} ms", flush=True)
if tot:
    a = np.array(tot)
    print(f"mean over decomposing generations: {a[:,0].mean():.3f} ms, sweeps mean {a[:,1].mean():.2f} (min {a[:,1].min():.0f}, max {a[:,1].max():.0f}); launched max_sweeps {ms}")
