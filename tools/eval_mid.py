"""sx_eval on rows of run-time length between the compile-time shapes (257 ... 2048 elements, and a few short ones): us per launch and
the fraction of the HBM peak on (8n + 8) B per evaluation.  Usage: python tools/eval_mid.py  (A/B: tools/ab_lib.py <lib.so> tools/eval_mid.py)"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from stochopy_amd import _device, _lib

ctx = _device.Context()
print("library:", _lib.LIB_PATH, flush=True)
SHORT = len(sys.argv) > 1 and sys.argv[1] == "short"
GRID = len(sys.argv) > 1 and sys.argv[1] == "grid"
LONG = len(sys.argv) > 1 and sys.argv[1] == "long"
for name in ("rosenbrock", "sphere", "rastrigin", "ackley", "griewank", "quartic", "styblinski_tang"):
    for n in ((512, 1024, 2048, 2049, 3000, 4096) if LONG else (64, 128, 256) if GRID else (20, 50, 100, 130, 200, 250, 256) if SHORT else (100, 200, 300, 500, 700, 1000, 1500, 2000)):
        P = ((1 << 27) // n) // 64 * 64 + (8 if SHORT else 0)
        X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
        f = ctx.empty((P,))
        torch.cuda.synchronize()
        with torch.cuda.stream(ctx.stream):
            for _ in range(3):
                _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            for _ in range(20):
                _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
            e1.record(ctx.stream); ctx.sync()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"sx_eval {name:10s} n={n:5d} P={P:8d}: {us:8.1f} us  {(8*n+8)*P/us/1e3/8000:.2f} of 8 TB/s", flush=True)
        del X
