"""Debug helper (round 6): build a -DSX_EIGH_TRACE variant of the library, run one cold decomposition through the resident kernel
and print the wall-clock timeline (100 MHz stamps, one clock for the whole chip) of the pair workgroups and of the first tile
workers over a few rounds.  usage: trace_eigh_flow.py [n round0]"""
import ctypes as C, glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
# build (here, where hipcc is): the traced solver linked with the other objects of the product library
#   hipcc ... -DSX_EIGH_TRACE -c stochopy_amd/csrc/sx_eigh.hip -o build_ab/sx_eigh_trace.o
#   hipcc --offload-arch=gfx950 -shared -fPIC build_ab/sx_eigh_trace.o <stochopy_amd/lib/*.o without sx_eigh.o> -o build_ab/libsx_etrace.so
out = os.path.join(ROOT, "build_ab", "libsx_etrace.so")
if not os.path.exists(out):
    os.makedirs(os.path.dirname(out), exist_ok=True)
    obj = os.path.join(ROOT, "build_ab", "sx_eigh_trace.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-DSX_EIGH_TRACE",
                    "-c", os.path.join(ROOT, "stochopy_amd", "csrc", "sx_eigh.hip"), "-o", obj], check=True)
    others = [f for f in sorted(glob.glob(os.path.join(ROOT, "stochopy_amd", "lib", "*.o"))) if not f.endswith("sx_eigh.o")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj] + others + ["-o", out], check=True)
from stochopy_amd import _lib
_lib.LIB_PATH = out
_lib.PROTOTYPES["sx_eigh_ftrace_read"] = (C.c_int, [C.c_void_p])
_lib.PROTOTYPES["sx_eigh_ftrace_set_round"] = (C.c_int, [C.c_int])
from stochopy_amd import _device
from stochopy_amd.linalg import Eigh
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
r0 = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ctx = _device.Context()
L = ctx.L
rs = np.random.RandomState(0)
A = rs.randn(n, n); Cm = A @ A.T / n + 0.1 * np.eye(n)
eig = Eigh(ctx, n)
assert L.sx_eigh_ftrace_set_round(r0) == 0
for rep in range(2):
    eig(ctx.upload(Cm), max_sweeps=4)   # not converged: every round sweeps
    ctx.sync()
buf = np.zeros(64 * 8 * 16, dtype=np.uint64)
L.sx_eigh_ftrace_read(buf.ctypes.data)
b = buf.reshape(64, 8, 16).astype(np.int64)
np_ = n // 32
t0 = b[:np_, 0, 0].min()
us = lambda x: (x - t0) / 100.0
print("pair workgroups, rounds %d..%d: us since the first pair staged round %d" % (r0, r0 + 7, r0))
print("slots: 5 staging begins | 6 staged | 7 pivot formed | 8 sweep done | 9 rotation stored (tagged halves, then plain) | 2 next round's loads issued | 10 drained | 11 counted | 3 predecessors' rotations here")
t0 = b[:np_, 0, 5].min()
for r in range(6):
    print(" round", r0 + r)
    for g in (0, 1, 5, 15):
        print("   pair %2d: " % g + " ".join("%7.2f" % us(b[g, r, s]) for s in (5, 6, 7, 8, 9, 2, 10, 11, 3)))
    d = b[:np_, r, :]
    seg = [("stage", 5, 6), ("pivot products", 6, 7), ("sweep", 7, 8), ("store U", 8, 9), ("issue next loads", 9, 2), ("drain", 2, 10), ("count", 10, 11), ("rotations here", 11, 3)]
    print("   mean over pairs: " + " | ".join("%s %.2f" % (nm, ((d[:, k1] - d[:, k0]) / 100.0).mean()) for nm, k0, k1 in seg)
          + " | -> next staging %.2f | round %.2f" % (((b[:np_, r + 1, 5] - d[:, 3]) / 100.0).mean(), ((b[:np_, r + 1, 5] - b[:np_, r, 5]) / 100.0).mean()))
print("tile workers (workgroups %d..63): slots 0 entry | 1 tiles of k-1 counted | 2 all rotations of k-1 | 4 first tile staged | 5 M tile done | 6 V tile done | 7 drained | 8 counted" % np_)
for r in range(4):
    print(" round", r0 + r)
    for g in (np_, np_ + 1, np_ + 20, 63):
        print("   worker %2d: " % (g - np_) + " ".join("%7.2f" % us(b[g, r, s]) for s in (0, 1, 2, 4, 5, 6, 7, 8)))
    d = b[np_:, r, :]
    seg = [("wait tiles", 0, 1), ("wait rotations", 1, 2), ("loads -> staged", 2, 4), ("M tile", 4, 5), ("V tile", 5, 6), ("drain", 6, 7), ("count", 7, 8)]
    print("   mean over workers: " + " | ".join("%s %.2f" % (nm, ((d[:, k1] - d[:, k0]) / 100.0).mean()) for nm, k0, k1 in seg))
