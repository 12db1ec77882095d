"""sx_eval at moderate population sizes: from which P on do the eight-lanes-per-row kernels (eval_r8_rt / eval_r8_long) beat the
16 / 32 / 64-lanes-per-row kernel?  Run once per SX_EVAL_R8_MIN (0: always eight lanes per row; 1000000000: never)."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from stochopy_amd import _device, _lib

ctx = _device.Context()
print("SX_EVAL_R8_MIN =", os.environ.get("SX_EVAL_R8_MIN"), flush=True)
for name in ("rosenbrock", "ackley"):
    for n in (100, 200, 300, 1000, 2000):
        for P in (1024, 2048, 4096, 8192, 16384, 32768):
            X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
            f = ctx.empty((P,))
            torch.cuda.synchronize()
            with torch.cuda.stream(ctx.stream):
                for _ in range(5):
                    _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(ctx.stream)
                for _ in range(200):
                    _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
                e1.record(ctx.stream); ctx.sync()
            print(f"sx_eval {name:10s} n={n:5d} P={P:6d}: {e0.elapsed_time(e1) / 200 * 1e3:8.2f} us", flush=True)
