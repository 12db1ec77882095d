"""Host timeline of a whole wide VD-CMA minimize() call: every allocation / upload / D2H read / library call, with its offset."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import stochopy_amd as sa
from stochopy_amd import _lib, _device
n, P, G = 16384, 1024, 200
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, sigma=0.3)
run = lambda m: sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma", options=dict(o, maxiter=m))
run(10); run(G)
ev = []
T0 = [0.0]
def wrap(obj, name, label):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); ev.append((label, t0 - T0[0], time.perf_counter() - t0)); return r
    setattr(obj, name, g)
for nm in ("empty", "zeros", "upload", "upload_async", "sync"):
    wrap(_device.Context, nm, "ctx." + nm)
wrap(torch.Tensor, "cpu", "Tensor.cpu")
L = _lib.lib()
orig = L.sx_vdcma_generation
class W:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); ev.append(("sx_vdcma_generation", t0 - T0[0], time.perf_counter() - t0)); return r
L.sx_vdcma_generation = W()
torch.cuda.synchronize(); T0[0] = time.perf_counter(); r = run(G); t_ret = time.perf_counter() - T0[0]; torch.cuda.synchronize()
print(f"minimize returned after {t_ret*1e3:.1f} ms, drained after {(time.perf_counter()-T0[0])*1e3:.1f} ms")
big = [e for e in ev if e[2] > 3e-4]
for lab, off, d in big[:60]:
    print(f"  +{off*1e3:8.2f} ms  {d*1e3:8.2f} ms  {lab}")
first = next(e for e in ev if e[0] == "sx_vdcma_generation"); last = [e for e in ev if e[0] == "sx_vdcma_generation"][-1]
print(f"first generation call at +{first[1]*1e3:.2f} ms, last at +{last[1]*1e3:.2f} ms")
