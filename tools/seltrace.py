"""Debug: phase timestamps of the CPSO restart selection kernel (-DSX_SELTRACE build; scribbles on candfit[0..7])."""
import glob, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
src = os.path.join(ROOT, "stochopy_amd", "csrc")
out = "/tmp/libsx_seltrace.so"
pre = os.path.join(ROOT, "build_ab", "libsx_seltrace.so")  # prebuilt in the build container (travels with the snapshot)
if os.path.exists(pre):
    out = pre
else:
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-DSX_SELTRACE",
                    "-shared", "-x", "hip"] + sorted(glob.glob(src + "/*.hip") + glob.glob(src + "/*.cpp")) + ["-o", out], check=True)
from stochopy_amd import _lib
_lib.LIB_PATH = out
import torch
import stochopy_amd as sa
from stochopy_amd.optimize import _cpso
runs = []
orig = _cpso._PsoRun.__init__
def spy(self, *a, **k):
    runs.append(self); orig(self, *a, **k)
_cpso._PsoRun.__init__ = spy
stamps = []
orig_restart = _cpso._PsoRun._restart_device
def restart(self):
    orig_restart(self)
    self.ctx.sync()
    stamps.append(self.candfit[:8].cpu().numpy().view(np.uint64).astype(np.int64))
_cpso._PsoRun._restart_device = restart
os.environ["SX_NO_GRAPH"] = "1"
r = sa.optimize.minimize(sa.factory.ackley, [[-5.12, 5.12]] * 256, method="cpso",
                         options={"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": int(os.environ.get("SELTRACE_MAXITER", "30")), "return_all": True, "verbosity": 0.0, "updating": "deferred"})
step = int(os.environ.get("SELTRACE_EVERY", "1"))
for s in stamps[3::step][:40]:
    d = np.diff(s)
    print("ticks(10ns) between stamps 0..7 [state+radii | keys | minmax | hist | scan+digit | gather | rank]:", d.tolist())
