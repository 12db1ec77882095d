"""How often can the CPSO restart question (radius < delta, cpso/_cpso.py:410-412) be settled from the radius against the OLD
best (a by-product of the generation kernel) and the step of the best?  One C3b run (Ackley n=256, P=16384, Philox), generation
by generation, torch arithmetic on the device beside the engine's own kernels:  python tools/cpso_radius_decisions.py [maxiter]"""
import os, sys
os.environ["SX_NO_GRAPH"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import stochopy_amd as sa
from stochopy_amd.optimize import _cpso

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
n = 256
cnt = {"known": 0, "above": 0, "below": 0, "exact": 0}
log = []
orig_gen = _cpso._PsoRun._generation


def gen(self):
    g0 = self.gbest.clone()
    orig_gen(self)
    self.ctx.sync()
    X, g1 = self.X, self.gbest
    thr = float(self.delta) * np.sqrt(4.0 * n)
    r = float(torch.sqrt(((X - g0) ** 2).sum(1)).max())
    d = float(torch.sqrt(((g1 - g0) ** 2).sum()))
    R = float(torch.sqrt(((X - g1) ** 2).sum(1)).max())
    k = "known" if d == 0.0 else "above" if r - d > thr * (1 + 1e-6) else "below" if r + d < thr * (1 - 1e-6) else "exact"
    cnt[k] += 1
    log.append((k, r / thr, d / thr, R / thr))


_cpso._PsoRun._generation = gen
_cpso._PsoRun.CHECK_EVERY = 1
r = sa.optimize.minimize(sa.factory.ackley, [[-5.12, 5.12]] * n, method="cpso",
                         options={"popsize": 16384, "seed": 0, "rng": "philox", "ftol": -1.0, "xtol": 0.0, "maxiter": m,
                                  "updating": "deferred", "return_all": True, "verbosity": 0.0})
print("generations", len(log), cnt, "nit", r.nit)
below = sum(1 for l in log if l[3] < 1.0)
print("generations with radius < delta (a restart is due if nw > 0):", below)
ex = [(i, round(l[1], 5), round(l[2], 5), round(l[3], 5)) for i, l in enumerate(log) if l[0] == "exact"]
print("exact-needed (generation, r/thr, d/thr, R/thr):", ex[:30])
for i in range(0, len(log), 200):
    seg = log[i:i + 200]
    print(i, {k: sum(1 for s in seg if s[0] == k) for k in cnt}, "R/thr", round(min(s[3] for s in seg), 3), "...", round(max(s[3] for s in seg), 3),
          "median d/thr", float(np.median([s[2] for s in seg])))
