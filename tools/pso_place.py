"""Does the placement of X / V / pbest in HBM matter to the PSO generation kernel?  (C-ABI level, HIP events)

    python tools/pso_place.py [objective] [n] [P]

X, V and pbest are carved out of one buffer `skew` doubles further apart than their size; each placement is timed as
`reps` replays of a 50-generation graph (events on the engine stream), the minimum and the median are printed.
"""
import ctypes as C
import sys

sys.path.insert(0, "/root/repo")
import numpy as np
import torch

import stochopy_amd as sa
from stochopy_amd import _device, _lib, _rng

name = sys.argv[1] if len(sys.argv) > 1 else "sphere"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
P = int(sys.argv[3]) if len(sys.argv) > 3 else 16384
skews = [int(s) for s in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 32, 512, 8192, 8192 + 512, 65536 + 32, 1 << 20, (1 << 20) + 4096 + 32]
NGEN, REPS = 50, 12

ctx = _device.Context()
L, ptr = ctx.L, _device.ptr
fid = getattr(sa.factory, name).sx_id
rng = np.random.default_rng(0)
X0 = rng.uniform(-5.12, 5.12, (P, n))
with torch.cuda.stream(ctx.stream):
    big = ctx.empty((3 * P * n + 3 * max(skews) + 64,))
    for skew in skews:
        X = big[0 : P * n].view(P, n)
        V = big[P * n + skew : 2 * P * n + skew].view(P, n)
        pbest = big[2 * P * n + 2 * skew : 3 * P * n + 2 * skew].view(P, n)
        X.copy_(torch.from_numpy(X0))
        V.zero_()
        pbest.copy_(X)
        npart = int(L.sx_num_partials(P, n))
        pbestfit, candfit = ctx.empty((P,)), ctx.empty((P,))
        part_f, part_i = ctx.empty((npart,)), ctx.empty((npart,), dtype=torch.int64)
        _lib.check(L.sx_eval(fid, ptr(X), P, n, n, None, None, ptr(pbestfit), None, None, ctx.stream_ptr), "sx_eval")
        g = int(pbestfit.argmin())
        gbest = X[g].clone()
        st = _lib.SxState(it=1, gbidx=g, gfit=float(pbestfit[g]), dx=0.0, status=_lib.SX_STATUS_NONE, done=0)
        state = ctx.upload(np.frombuffer(bytes(st), dtype=np.int64).copy())
        lower, upper = ctx.upload(np.full(n, -5.12)), ctx.upload(np.full(n, 5.12))
        a = _lib.SxPsoArgs()
        a.X, a.V, a.pbest = X.data_ptr(), V.data_ptr(), pbest.data_ptr()
        a.pbestfit, a.candfit, a.gbest = pbestfit.data_ptr(), candfit.data_ptr(), gbest.data_ptr()
        a.lower, a.upper, a.state = lower.data_ptr(), upper.data_ptr(), state.data_ptr()
        a.part_f, a.part_i = part_f.data_ptr(), part_i.data_ptr()
        a.P, a.ld, a.row0, a.n = P, n, 0, n
        a.fun_id, a.constraints, a.rng, a.maxiter = fid, 0, _lib.SX_RNG_PHILOX, 1 << 30
        a.w, a.c1, a.c2, a.xtol, a.ftol = 0.73, 1.496, 1.496, 0.0, -1.0
        a.key0, a.key1 = _rng.philox_key(0)
        graph = C.c_void_p()
        _lib.check(L.sx_pso_graph_create(C.byref(a), NGEN, None, 0.0, 0.0, None, C.byref(graph)), "graph")
        times = []
        for r in range(REPS + 2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(ctx.stream)
            _lib.check(L.sx_graph_launch(graph, ctx.stream_ptr), "launch")
            e1.record(ctx.stream)
            e1.synchronize()
            if r >= 2:
                times.append(e0.elapsed_time(e1) * 1e3 / NGEN)
        L.sx_graph_destroy(graph)
        times.sort()
        byts = (48 * n + 24) * P
        print(f"pso {name} n={n} P={P} skew={skew:8d} doubles ({(skew * 8) % 4096:5d} B mod 4K): min {times[0]:6.1f} "
              f"median {times[len(times) // 2]:6.1f} max {times[-1]:6.1f} us/gen (generation + finalise)  "
              f"-> {byts / times[0] / 1e3:7.1f} GB/s at min", flush=True)
