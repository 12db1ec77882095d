"""Host time of every sx_vdcma_generation call inside a whole minimize() run (wide VD-CMA): which calls block?"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import stochopy_amd as sa
from stochopy_amd import _lib
n, P, G = 16384, 1024, 200
o = dict(seed=0, rng="philox", ftol=-1.0, xtol=0.0, backend="hip", popsize=P, sigma=0.3)
run = lambda m: sa.optimize.minimize(sa.factory.rosenbrock, [[-5.12, 5.12]] * n, method="vdcma", options=dict(o, maxiter=m))
run(10)
L = _lib.lib()
orig = L.sx_vdcma_generation
times = []
class Wrap:
    def __call__(self, *a):
        t0 = time.perf_counter(); r = orig(*a); times.append(time.perf_counter() - t0); return r
L.sx_vdcma_generation = Wrap()
torch.cuda.synchronize(); t0 = time.perf_counter(); r = run(G); torch.cuda.synchronize(); wall = time.perf_counter() - t0
t = np.array(times) * 1e6
print(f"wall {wall*1e3:.1f} ms for {r.nit} generations; calls: sum {t.sum()/1e3:.1f} ms, median {np.median(t):.0f} us, max {t.max():.0f} us")
print("calls slower than 150 us at generations:", [int(i) + 1 for i in np.nonzero(t > 150)[0]][:80])
print("first 24:", [int(x) for x in t[:24]])
