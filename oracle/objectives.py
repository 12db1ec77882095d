"""Batched restatement of the seven benchmark objectives (oracle, numpy).

Reference: stochopy/factory/benchmark.py:14-156 (scalar functions of one
(n,) vector, called popsize times per generation through the population
wrapper stochopy/optimize/_common.py:79-80).  Here each objective maps a
(P, n) float64 array to (P,) values in ONE vectorised numpy expression whose
per-row arithmetic (operation order, pairwise `add.reduce` over the contiguous
last axis) is identical to the reference's per-row call -- pinned bit-for-bit
by tests/golden/factory_kat.json.
"""
import numpy as np

TWO_PI = 2.0 * np.pi


def _rows(X):
    X = np.ascontiguousarray(X, dtype=np.float64)
    if X.ndim == 1:
        X = X[None, :]
    return X


def ackley(X):
    """benchmark.py:14-34.  e is the literal 2.7182818284590451 (:31)."""
    X = _rows(X)
    n = X.shape[1]
    e = 2.7182818284590451
    s1 = np.sqrt(1.0 / n * np.square(X).sum(axis=1))
    s2 = 1.0 / n * np.cos(TWO_PI * X).sum(axis=1)
    return 20.0 + e - 20.0 * np.exp(-0.2 * s1) - np.exp(s2)


def griewank(X):
    """benchmark.py:37-56."""
    X = _rows(X)
    n = X.shape[1]
    s1 = np.square(X).sum(axis=1) / 4000.0
    p1 = np.prod(np.cos(X / np.sqrt(np.arange(1, n + 1))), axis=1)
    return 1.0 + s1 - p1


def quartic(X):
    """benchmark.py:59-76."""
    X = _rows(X)
    n = X.shape[1]
    return (np.arange(1, n + 1) * np.power(X, 4)).sum(axis=1)


def rastrigin(X):
    """benchmark.py:79-97."""
    X = _rows(X)
    n = X.shape[1]
    s1 = (np.square(X) - 10.0 * np.cos(TWO_PI * X)).sum(axis=1)
    return 10.0 * n + s1


def rosenbrock(X):
    """benchmark.py:100-118: two separate sums of length n-1, then 100*s1+s2."""
    X = _rows(X)
    head = X[:, :-1]
    s1 = ((X[:, 1:] - head**2) ** 2).sum(axis=1)
    s2 = np.square(1.0 - head).sum(axis=1)
    return 100.0 * s1 + s2


def sphere(X):
    """benchmark.py:121-136."""
    X = _rows(X)
    return np.square(X).sum(axis=1)


def styblinski_tang(X):
    """benchmark.py:139-156 (constant 39.16599 at :156)."""
    X = _rows(X)
    n = X.shape[1]
    s1 = (np.power(X, 4) - 16.0 * np.square(X) + 5.0 * X).sum(axis=1)
    return 0.5 * s1 + 39.16599 * n


OBJECTIVES = {
    "ackley": ackley,
    "griewank": griewank,
    "quartic": quartic,
    "rastrigin": rastrigin,
    "rosenbrock": rosenbrock,
    "sphere": sphere,
    "styblinski_tang": styblinski_tang,
}


def evaluate(name, X):
    """Population wrapper of stochopy/optimize/_common.py:79-80 for a named objective."""
    return OBJECTIVES[name](X)


# --------------------------------------------------------------------------- #
# numpy's pairwise add.reduce order, restated in pure Python.
# Used by tests to pin WHY the device kernels' summation order is what it is;
# small inputs only.
# --------------------------------------------------------------------------- #
def pairwise_sum_py(a):
    """0 + PW(a): numpy/_core/src/umath/loops_utils.h.src pairwise sum, blocksize 128."""

    def pw(lo, m):
        if m < 8:
            res = 0.0
            for i in range(m):
                res = res + a[lo + i]
            return res
        if m <= 128:
            r = [a[lo + j] for j in range(8)]
            i = 8
            while i < m - (m % 8):
                for j in range(8):
                    r[j] = r[j] + a[lo + i + j]
                i += 8
            res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]))
            while i < m:
                res = res + a[lo + i]
                i += 1
            return res
        h = m // 2
        h -= h % 8
        return pw(lo, h) + pw(lo + h, m - h)

    a = [float(v) for v in a]
    return 0.0 + pw(0, len(a))
