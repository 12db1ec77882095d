"""Debug helper: build a -DSX_TRACE variant of the library, run a few DE generations, print checkpoint deltas."""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "stochopy_amd", "csrc")
out = "/tmp/libsx_trace.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-DSX_TRACE",
                "-shared", "-x", "hip"] + sorted(glob.glob(src + "/*.hip") + glob.glob(src + "/*.cpp")) + ["-o", out], check=True)
from stochopy_amd import _lib
_lib.LIB_PATH = out
_lib.PROTOTYPES["sx_trace_read"] = (C.c_int, [C.c_void_p])
import torch
from stochopy_amd.optimize import _de
n, P = int(sys.argv[1]) if len(sys.argv) > 1 else 128, int(sys.argv[2]) if len(sys.argv) > 2 else 4096
run = _de._DeRun(_lib.FUN_IDS["rosenbrock"], np.full(n, -5.12), np.full(n, 5.12), None, 10**6, P, 0.5, 0.9, "best1bin", None,
                 0.0, -1.0, False, 1.0, None, "philox", 1, 1, autorun=False)
with torch.cuda.stream(run.ctx.stream):
    run._setup()
    run.enqueue(30)
    run.ctx.sync()
    buf = np.zeros(1024 * 16, dtype=np.uint64)
    run.ctx.L.sx_trace_read(buf.ctypes.data)
full = buf.reshape(1024, 16).astype(np.int64)
b = full[:, :8]
clk = full[:, 8:]
print('chain', run.chain)
nb = min(1024, int(run.ctx.L.sx_num_partials(P, n)))
b = b[:nb]
t0 = b[:, 0].min()
print("blocks", nb, "units: 10ns ticks (100MHz)")
print("block start spread: max(start)-min(start) =", b[:, 0].max() - t0)
print("kernel span (min start -> max end) =", b[:, 5].max() - t0)
print("0 -> 6 (state s_load):", (b[:, 6] - b[:, 0]).mean())
print("6 -> 1 (donors):", (b[:, 1] - b[:, 6]).mean())
for k in range(2, 6):
    d = b[:, k] - b[:, k - 1]
    print(f"phase {k-1}->{k}: mean {d.mean():.1f} min {d.min()} max {d.max()}")
cyc = (clk[:nb, 5] - clk[:nb, 0]).mean(); wall = (b[:, 5] - b[:, 0]).mean()
print("shader cycles per block %.0f over %.0f ticks -> %.2f GHz" % (cyc, wall, cyc / (wall * 10.0)))
print("per-block total: mean", (b[:, 5] - b[:, 0]).mean(), "max", (b[:, 5] - b[:, 0]).max())
