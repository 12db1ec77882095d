"""FETCH_SIZE calibration on a known byte count with the row kernels' own access pattern
(8-byte lanes, rows of n doubles): sx_eval reads exactly P*n*8 bytes."""
import sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import stochopy_amd as sa
from stochopy_amd import _device, _lib
ctx = _device.Context()
for (P, n) in ((262144, 128), (32768, 1024)):
    X = torch.rand(P, n, dtype=torch.float64, device="cuda")
    f = torch.empty(P, dtype=torch.float64, device="cuda")
    with torch.cuda.stream(ctx.stream):
        for _ in range(5):
            _device.evaluate(ctx, _lib.FUN_IDS["sphere"], X, n, f=f)
        ctx.sync()
    print("calib", P, n, "bytes", P * n * 8)
