"""Objective-only evaluation (sx_eval): evals/s and GB/s against the 8n+8 bytes per evaluation of SURVEY.md 8(d)."""
import sys
sys.path.insert(0, "/root/repo")
import torch
from stochopy_amd import _device, _lib

ctx = _device.Context()
for name, n, P in (("rosenbrock", 128, 4096), ("rosenbrock", 128, 1 << 20), ("rastrigin", 128, 1 << 20),
                   ("rosenbrock", 1024, 1 << 17), ("ackley", 256, 1 << 19), ("sphere", 64, 1 << 21),
                   ("rosenbrock", 512, 1 << 18), ("rosenbrock", 2048, 1 << 16), ("ackley", 1024, 1 << 17),
                   ("rastrigin", 1024, 1 << 17), ("sphere", 1024, 1 << 17), ("rosenbrock", 1024, 1 << 13)):
    X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
    f = ctx.empty((P,))
    fid = _lib.FUN_IDS[name]
    with torch.cuda.stream(ctx.stream):
        for _ in range(5):
            _device.evaluate(ctx, fid, X, n, f=f)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record(ctx.stream)
        for _ in range(reps):
            _device.evaluate(ctx, fid, X, n, f=f)
        e1.record(ctx.stream)
        ctx.sync()
    us = e0.elapsed_time(e1) / reps * 1e3
    byts = (8 * n + 8) * P
    print(f"sx_eval {name:11s} n={n:5d} P={P:8d}: {us:9.1f} us  {P/us*1e6:.3e} evals/s  {byts/us/1e3:8.1f} GB/s "
          f"({byts/us/1e3/8000:.2f} of 8 TB/s)")
