"""Debug helper: build a -DSX_EIGH_TRACE variant of the library, run one decomposition, print the shader-clock
phases of the pair workgroups of the round kernel (csrc/sx_eigh.hip).  usage: trace_eigh.py [n]"""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "stochopy_amd", "csrc")
out = "/tmp/libsx_etrace.so"
subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-DSX_EIGH_TRACE",
                "-shared", "-x", "hip"] + sorted(glob.glob(src + "/*.hip") + glob.glob(src + "/*.cpp")) + ["-o", out], check=True)
from stochopy_amd import _lib
_lib.LIB_PATH = out
_lib.PROTOTYPES["sx_eigh_trace_read"] = (C.c_int, [C.c_void_p])
from stochopy_amd import _device
from stochopy_amd.linalg import Eigh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
ctx = _device.Context()
rs = np.random.RandomState(0)
A = rs.randn(n, n); Cm = A @ A.T / n + 0.1 * np.eye(n)
eig = Eigh(ctx, n)
eig(ctx.upload(Cm), max_sweeps=3)   # not converged: every launch sweeps
ctx.sync()
buf = np.zeros(64 * 16, dtype=np.uint64)
ctx.L.sx_eigh_trace_read(buf.ctypes.data)
b = buf.reshape(64, 16).astype(np.int64)[: max(1, n // 32)]
for nm, k0, k1 in (("load tiles+U", 0, 1), ("pivot (MFMA) + off2", 1, 3), ("sweep", 3, 4), ("store U", 4, 5)):
    d = b[:, k1] - b[:, k0]
    print(f"{nm:20s} mean {d.mean():9.0f} cycles  min {d.min()} max {d.max()}")
print("pivot phase, wave 0: first products (LDS reads + 8 MFMAs + Y store)", (b[:, 2] - b[:, 1]).mean(), "barrier", (b[:, 9] - b[:, 2]).mean(),
      "second products + pivot store", (b[:, 15] - b[:, 9]).mean(), "barrier", (b[:, 3] - b[:, 15]).mean())
print("total", (b[:, 5] - b[:, 0]).mean(), "cycles (the last launch is a cross-only round: 16 inner rounds); per inner round",
      (b[:, 4] - b[:, 3]).mean() / 16)
print("inner round 5, updating wave 0: work", (b[:, 7] - b[:, 6]).mean(), "barrier wait", (b[:, 8] - b[:, 7]).mean())
print("inner round 5, rotation lane 0: loads+pivot", (b[:, 11] - b[:, 10]).mean(), "rotation", (b[:, 12] - b[:, 11]).mean(),
      "store", (b[:, 13] - b[:, 12]).mean(), "barrier wait", (b[:, 14] - b[:, 13]).mean())
print("start skew duty-vs-wave0", (b[:, 10] - b[:, 6]).mean(), "end skew", (b[:, 14] - b[:, 8]).mean())
