"""One wide-row sx_eval shape, launched a number of times (for rocprofv3 --pmc passes): python tools/wide_one.py fun n P [launches]"""
import sys
sys.path.insert(0, "/root/repo")
import torch
from stochopy_amd import _device, _lib

name, n, P = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
ctx = _device.Context()
X = torch.rand((P, n), dtype=torch.float64, device=ctx.device) * 10.24 - 5.12
f = ctx.empty((P,))
with torch.cuda.stream(ctx.stream):
    for _ in range(reps):
        _device.evaluate(ctx, _lib.FUN_IDS[name], X, n, f=f)
    ctx.sync()
print("done", name, n, P, reps)
