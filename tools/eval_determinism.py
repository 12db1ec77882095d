"""Is sx_eval deterministic from call to call, and equal to the oracle, at large P?  (found while A/B-ing the eval kernels: the
order-free hash of Ackley n=128 P=2^20 differed between processes)"""
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import oracle
from stochopy_amd import _device, _lib

ctx = _device.Context()
for name, n, P in (("ackley", 128, 1 << 20), ("ackley", 128, 1 << 16), ("ackley", 64, 1 << 18), ("ackley", 256, 1 << 18),
                   ("rastrigin", 128, 1 << 18), ("griewank", 128, 1 << 18), ("rosenbrock", 128, 1 << 18), ("rosenbrock", 256, 1 << 16),
                   ("sphere", 64, 1 << 18)):
    g = torch.Generator(device=ctx.device).manual_seed(1)
    X = torch.rand((P, n), dtype=torch.float64, device=ctx.device, generator=g) * 10.24 - 5.12
    torch.cuda.synchronize()  # X is produced on torch's default stream, the evaluation runs on the engine stream
    fid = _lib.FUN_IDS[name]
    outs = []
    with torch.cuda.stream(ctx.stream):
        for k in range(4):
            f = torch.full((P,), float(k + 1) * 1e300, dtype=torch.float64, device=ctx.device)  # sentinels: an unwritten row shows
            torch.cuda.synchronize()
            _device.evaluate(ctx, fid, X, n, f=f)
            ctx.sync()
            outs.append(f.cpu().numpy())
    same = [int((outs[0].view(np.uint64) != o.view(np.uint64)).sum()) for o in outs[1:]]
    Xh = X.cpu().numpy()
    ref = np.concatenate([oracle.OBJECTIVES[name](Xh[k : k + (1 << 16)]) for k in range(0, P, 1 << 16)])
    got = outs[0]
    nbit = int((ref.view(np.uint64) != got.view(np.uint64)).sum())
    rel = float(np.max(np.abs(ref - got) / np.maximum(1e-300, np.abs(ref))))
    print(f"{name:10s} n={n:4d} P={P:8d}: rows that differ between calls {same}; vs oracle (all rows): {nbit} rows not bit-equal, "
          f"max rel {rel:.2e}", flush=True)
