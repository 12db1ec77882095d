#!/usr/bin/env python3
"""Which symmetric eigensolver for the CMA-ES model update (cmaes/_cmaes.py:301-309) on the GPU?
Times torch.linalg.eigh against rocSOLVER's syevd / syevdj / syevj (called directly, ctypes) on a
covariance-like fp64 matrix, and checks residual + orthogonality of each.  usage: bench_eigh.py [n]"""
import ctypes as C
import sys
import time

import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dev = torch.device("cuda", 0)
torch.manual_seed(0)
# a CMA-like covariance: identity plus a few hundred rank-one updates
A = torch.eye(n, dtype=torch.float64, device=dev)
for _ in range(8):
    Y = torch.randn(n, n, dtype=torch.float64, device=dev) * 0.05
    A = 0.9 * A + Y.T @ Y / n
A = 0.5 * (A + A.T)

rb = C.CDLL("librocblas.so")
rs = C.CDLL("librocsolver.so")
h = C.c_void_p()
assert rb.rocblas_create_handle(C.byref(h)) == 0
rb.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
EVECT, UPPER, ASC = 211, 121, 252
info = torch.zeros(1, dtype=torch.int32, device=dev)
W = torch.zeros(n, dtype=torch.float64, device=dev)
E = torch.zeros(n, dtype=torch.float64, device=dev)
resid = torch.zeros(1, dtype=torch.float64, device=dev)
nsw = torch.zeros(1, dtype=torch.int32, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())


def syevd(M):
    rc = rs.rocsolver_dsyevd(h, EVECT, UPPER, n, p(M), n, p(W), p(E), p(info))
    assert rc == 0, rc


def syevdj(M):
    rc = rs.rocsolver_dsyevdj(h, EVECT, UPPER, n, p(M), n, p(W), p(info))
    assert rc == 0, rc


def syevj(M):
    rc = rs.rocsolver_dsyevj(h, ASC, EVECT, UPPER, n, p(M), n, C.c_double(0.0), p(resid), 100, p(nsw), p(W), p(info))
    assert rc == 0, rc


def check(name, w, V):
    r = (A @ V - V * w).abs().max().item()
    o = (V.T @ V - torch.eye(n, dtype=torch.float64, device=dev)).abs().max().item()
    return f"resid {r:.2e} orth {o:.2e}"


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


out = {}
ms = timeit(lambda: torch.linalg.eigh(A))
w, V = torch.linalg.eigh(A)
print(f"n={n} torch.linalg.eigh   {ms:8.2f} ms  {check('torch', w, V)}", flush=True)
for name, fn in (("syevd", syevd), ("syevdj", syevdj), ("syevj", syevj)):
    try:
        M = A.clone()
        ms = timeit(lambda: (M.copy_(A), fn(M)))
        M.copy_(A)
        fn(M)
        torch.cuda.synchronize()
        # column-major eigenvectors in M == row-major M.T
        extra = f" sweeps {int(nsw.item())}" if name == "syevj" else ""
        print(f"n={n} rocsolver_d{name:7s} {ms:8.2f} ms  {check(name, W, M.T)} info {int(info.item())}{extra}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(name, "failed:", e, flush=True)
