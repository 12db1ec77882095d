#!/usr/bin/env python3
"""Kernel-level timings of the CMA-ES / VD-CMA pieces outside the eigensolver (GPU box), HIP events around 200 launches each:
  sample   sx_cmaes_sample      arx = xmean + sigma * (Z o D) B^T       2*P*n*n flops
  rank-mu  sx_cmaes_rank_mu     Y, then C = decay*C + cmu*Y^T diag(w) Y + c1*pc pc^T     2*n*n*mu flops
  vdsample sx_vdcma_sample      one pass over Z, two outputs            3*P*n*8 bytes
usage: bench_cma_kernels.py [n P]..."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stochopy_amd import _device, _lib

ctx = _device.Context()
p = _device.ptr
shapes = [(512, 1024), (256, 512), (128, 256), (1024, 2048)]
if len(sys.argv) > 2:
    a = list(map(int, sys.argv[1:]))
    shapes = list(zip(a[0::2], a[1::2]))
REP = 200


def timed(fn):
    with torch.cuda.stream(ctx.stream):
        for _ in range(10):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(ctx.stream)
        for _ in range(REP):
            fn()
        e1.record(ctx.stream)
        e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / REP  # us


for n, P in shapes:
    mu = P // 2
    rs = np.random.RandomState(n)
    d = {k: ctx.upload(v) for k, v in dict(Z=rs.randn(P, n), B=np.linalg.qr(rs.randn(n, n))[0], D=rs.uniform(0.5, 2, n), xm=rs.randn(n),
                                            arx=rs.randn(P, n), w=np.full(mu, 1.0 / mu), xold=rs.randn(n), pc=rs.randn(n),
                                            C=np.eye(n), dv=rs.uniform(0.5, 2, n), vn=rs.randn(n) / np.sqrt(n)).items()}
    idx = ctx.upload(np.ascontiguousarray(rs.permutation(P)[:mu], dtype=np.int64))
    out, Y, ary = ctx.empty((P, n)), ctx.empty((mu, n)), ctx.empty((P, n))
    t_s = timed(lambda: _lib.check(ctx.L.sx_cmaes_sample(p(d["xm"]), 0.3, p(d["B"]), p(d["D"]), p(d["Z"]), p(out), P, n, ctx.stream_ptr)))
    t_r = timed(lambda: _lib.check(ctx.L.sx_cmaes_rank_mu(p(d["arx"]), p(idx), p(d["w"]), mu, p(d["xold"]), 0.3, p(d["pc"]), 1e-3, 2e-3, 0.0,
                                                          p(d["C"]), p(Y), n, ctx.stream_ptr)))
    t_v = timed(lambda: _lib.check(ctx.L.sx_vdcma_sample(p(d["Z"]), P, n, 0, p(d["dv"]), p(d["vn"]), 0.7, p(d["xm"]), 0.3, None, p(ary), p(out),
                                                         ctx.stream_ptr)))
    fs, fr = 2.0 * P * n * n, 2.0 * n * n * mu
    print(f"n={n} P={P} mu={mu}: sample {t_s:7.2f} us {fs / t_s / 1e6:6.2f} TF ({fs / t_s / 1e6 / 78.6:.2f} of 78.6) | "
          f"rank-mu (Y + GEMM) {t_r:7.2f} us {fr / t_r / 1e6:6.2f} TF ({fr / t_r / 1e6 / 78.6:.2f}) | "
          f"vd sample {t_v:7.2f} us {3.0 * P * n * 8 / t_v / 1e6:5.2f} TB/s", flush=True)
