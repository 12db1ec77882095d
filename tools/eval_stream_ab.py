"""sx_eval on one-batch rows: eight lanes per row (csrc/sx_core.hip eval_r8_kernel) against the one-visit kernel with 16 / 32 / 64
lanes per row, same binary (SX_EVAL_R8 is read once per process: one process per arm).  The resident form with the loads
ahead that was tried first is recorded in profiles/r5_eval_stream_ab.txt (removed from the source).
Usage: python tools/eval_stream_ab.py            (spawns the arms)"""
import os
import subprocess
import sys

sys.path.insert(0, "/root/repo")
SHAPES = (("rosenbrock", 256, 1 << 19), ("rosenbrock", 128, 1 << 20), ("rastrigin", 128, 1 << 20), ("sphere", 64, 1 << 21), ("rosenbrock", 64, 1 << 21),
          ("ackley", 256, 1 << 19), ("sphere", 256, 1 << 19), ("rosenbrock", 128, 1 << 16), ("ackley", 128, 1 << 20))

if len(sys.argv) > 1 and sys.argv[1] == "arm":
    import numpy as np
    import torch

    from stochopy_amd import _device, _lib

    ctx = _device.Context()
    only = sys.argv[2].split(":") if len(sys.argv) > 2 else None  # e.g. rosenbrock:128:1048576 (counter passes: one kernel, one shape)
    for name, n, P in (SHAPES if only is None else ((only[0], int(only[1]), int(only[2])),)):
        g = torch.Generator(device=ctx.device).manual_seed(1)
        X = torch.rand((P, n), dtype=torch.float64, device=ctx.device, generator=g) * 10.24 - 5.12
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
        h = int(np.bitwise_xor.reduce(f.cpu().numpy().view(np.uint64)))  # all values, order-free: arms must agree bit for bit
        print(f"  sx_eval {name:11s} n={n:4d} P={P:8d}: {us:8.1f} us  {byts/us/1e3:7.1f} GB/s ({byts/us/1e3/8000:.3f})  xor {h:016x}",
              flush=True)
else:
    arms = [("one-visit kernel, 16 / 32 / 64 lanes per row (SX_EVAL_R8=0)", {"SX_EVAL_R8": "0"}), ("eight lanes per row (default)", {})]
    for label, env in arms:
        print(label, flush=True)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "arm"], env=dict(os.environ, **env), capture_output=True,
                             text=True)
        print("".join(ln + "\n" for ln in out.stdout.splitlines() if ln.startswith("  ")), end="", flush=True)
        if out.returncode:
            print(out.stderr[-2000:])
