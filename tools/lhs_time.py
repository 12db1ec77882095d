"""Host-side Latin hypercube of the initial population (P=4096, n=128): the strided column gather vs a gather along
contiguous rows of the transpose (same values)."""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from stochopy_amd import _rng
lo, hi = np.full(128, -5.12), np.full(128, 5.12)
P, n = 4096, 128
def strided(s):
    x = s.random((P, n)); x /= P; x += np.linspace(-1.0, 1.0, P, endpoint=False)[:, None]
    pop = np.empty((P, n))
    for j in range(n):
        pop[:, j] = x[s.permutation(P), j]
    pop *= 0.5 * (hi - lo); pop += 0.5 * (hi + lo); return pop
def transposed(s):
    x = s.random((P, n)); x /= P; x += np.linspace(-1.0, 1.0, P, endpoint=False)[:, None]
    xt = np.ascontiguousarray(x.T); popt = np.empty((n, P))
    for j in range(n):
        np.take(xt[j], s.permutation(P), out=popt[j])
    pop = np.ascontiguousarray(popt.T)
    pop *= 0.5 * (hi - lo); pop += 0.5 * (hi + lo); return pop
for rep in range(4):
    r = []
    for f in (strided, transposed):
        s = _rng.make_init_stream("philox", 5); t = time.perf_counter(); a = f(s); r.append((time.perf_counter() - t, a))
    s = _rng.make_init_stream("philox", 5); t = time.perf_counter(); s.random((P, n)); t3 = time.perf_counter() - t
    t = time.perf_counter()
    for j in range(n): s.permutation(P)
    t4 = time.perf_counter() - t
    print("strided %.2f ms  transposed %.2f ms | draws alone: random %.2f ms + 128 permutations %.2f ms  equal=%s"
          % (r[0][0] * 1e3, r[1][0] * 1e3, t3 * 1e3, t4 * 1e3, np.array_equal(r[0][1], r[1][1])))
