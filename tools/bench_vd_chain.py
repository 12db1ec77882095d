"""A/B of the wide VD-CMA generation: SX_VD_CHAIN from the environment (1: the model update's chain in one launch, 0: one
launch per phase); us per generation over device-resident runs at the sizes of the (f)-rows table."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import stochopy_amd as sa  # noqa: E402
from stochopy_amd import _device  # noqa: E402


def run(n, P, gens):
    opts = {"maxiter": gens, "popsize": P, "seed": 3, "sigma": 0.3, "backend": "hip", "rng": "philox", "xtol": 0.0, "ftol": -1.0}
    bounds = [[-3.0, 3.0]] * n
    sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="vdcma", options=dict(opts, maxiter=8))
    t = _device.torch()
    t.cuda.synchronize()
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        res = sa.optimize.minimize(sa.factory.rosenbrock, bounds, method="vdcma", options=opts)
        t.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return best / res.nit * 1e6, res


if __name__ == "__main__":
    print("SX_VD_CHAIN =", os.environ.get("SX_VD_CHAIN", "(default)"))
    for n, P, gens in [(16384, 1024, 200), (16384, 4096, 100), (65536, 512, 100), (8192, 1024, 200), (16384, 33, 200),
                       (262144, 64, 50)]:
        us, r = run(n, P, gens)
        print(f"VD-CMA rosenbrock n={n:6d} P={P:5d}: {us:8.1f} us/gen  {32.0 * n * P / us / 1e3 / 8000:.3f} of 8 TB/s on 32 n B "
              f"per candidate   fun {float(r.fun).hex()}", flush=True)
