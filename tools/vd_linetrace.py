"""Line-level host timeline of _VdDeviceRun.__init__ (wide VD-CMA): which source lines take more than 1 ms?"""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
import stochopy_amd as sa
from stochopy_amd.optimize import _vdcma
n, P, G = 16384, 1024, 200
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, sigma=0.3)
run = lambda m: sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma", options=dict(o, maxiter=m))
run(10); run(G)
code = _vdcma._VdDeviceRun.__init__.__code__
last = [None, 0.0]
slow = {}
def tracer(frame, event, arg):
    if frame.f_code is not code:
        return None
    def local(frame, event, arg):
        if event == "line":
            now = time.perf_counter()
            if last[0] is not None:
                d = now - last[1]
                if d > 1e-3:
                    slow.setdefault(last[0], []).append(d)
            last[0], last[1] = frame.f_lineno, now
        return local
    return local
for rep in range(2):
    slow.clear(); last[0] = None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sys.settrace(tracer); r = run(G); sys.settrace(None)
    torch.cuda.synchronize()
    print(f"call {rep}: {(time.perf_counter()-t0)*1e3:.1f} ms")
    for ln, ds in sorted(slow.items()):
        print(f"   line {ln}: {len(ds)} x, total {sum(ds)*1e3:.1f} ms, max {max(ds)*1e3:.1f} ms")
