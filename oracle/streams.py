"""Random draws of the hot path, in the order the reference consumes them (oracle).

Two providers with one interface:

* ``LegacyStream`` -- numpy's legacy global MT19937 stream, i.e. exactly what the
  reference draws after ``np.random.seed(seed)`` (de/_de.py:148-149,
  cpso/_cpso.py:153-154, cmaes/_cmaes.py:116-117).  It calls
  ``numpy.random.RandomState`` (the reference's own third-party dependency), in
  the reference's call order:
    LHS      _common.py:111-113   uniform(size=(P,n)) then n x permutation(P)
    DE       de/_de.py:250, 304-311, 340, de/_constraints.py:24
    PSO      cpso/_cpso.py:262-263, 422
    CMA-ES   cmaes/_cmaes.py:180, 234
* ``PhiloxStream`` -- the counter-based Philox4x32-10 layout the HIP kernels use in
  throughput mode (``rng="philox"``; DESIGN.md section "Philox layout").  Not a
  reference algorithm: it exists so CPU<->GPU parity can be checked at sizes the
  reference's O(P^2) donor permutations cannot reach.
"""
import numpy as np

# purposes (counter word 3)
PURPOSE_DE_CROSS = 0
PURPOSE_DE_DONOR = 1
PURPOSE_DE_RESAMPLE = 2
PURPOSE_PSO_R1 = 3
PURPOSE_PSO_R2 = 4
PURPOSE_PSO_RESTART = 5
PURPOSE_CMA_NORMAL = 6
PURPOSE_NA_UNIFORM = 7
PURPOSE_INIT_JITTER = 8
PURPOSE_INIT_PERM = 9

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = 0x9E3779B9
_W1 = 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)
_S32 = np.uint64(32)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al., SC'11) on broadcastable uint32 counter arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*[np.asarray(c, dtype=np.uint64) & _MASK for c in (c0, c1, c2, c3)])
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = (p1 >> _S32) ^ c1 ^ np.uint64(k0)
        n1 = p1 & _MASK
        n2 = (p0 >> _S32) ^ c3 ^ np.uint64(k1)
        n3 = p0 & _MASK
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u53(a, b):
    """numpy legacy double from two 32-bit words: ((a>>5)*2^26 + (b>>6)) / 2^53."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    return ((a >> np.uint64(5)).astype(np.float64) * 67108864.0 + (b >> np.uint64(6)).astype(np.float64)) / 9007199254740992.0


def _mulhi(w, m):
    return ((np.asarray(w, dtype=np.uint64) * np.uint64(m)) >> _S32).astype(np.int64)


class LegacyStream:
    """The reference's np.random.* stream (numpy legacy MT19937)."""

    kind = "numpy-legacy"

    def __init__(self, seed=None):
        self.rs = np.random.RandomState(seed) if seed is not None else np.random.RandomState()

    # -- initial population: _common.py:109-120 -------------------------------
    def lhs_draws(self, P, n):
        u = self.rs.uniform(size=(P, n))
        perms = [self.rs.permutation(P) for _ in range(n)]
        return u, perms

    # -- DE generation: de/_de.py:250 then :304-311 then :340 then constraints --
    def de_generation(self, gen, P, n, k, resample_bounds=None, row0=0):
        rs = self.rs
        r1 = rs.rand(P, n)
        donors = np.empty((k, P), dtype=np.int64)
        for i in range(P):
            # permutation(delete(arange(P), i)) == delete(...)[permutation(P-1)]: same draws
            p = rs.permutation(P - 1)[:k]
            donors[:, i] = p + (p >= i)
        irand = rs.randint(n, size=P)
        resample = None
        if resample_bounds is not None:
            lo, hi = resample_bounds
            resample = rs.uniform(lo, hi, (P, n))
        return {"r1": r1, "donors": donors, "irand": irand, "resample": resample}

    # -- DE generation, updating="immediate": de/_de.py:250, then PER INDIVIDUAL :376 (donor permutation),
    #    :380 (randint(ndim)) and the constraint's uniform(lower, upper, (n,)) (de/_constraints.py:24) --------
    def de_generation_async(self, gen, P, n, k, resample_bounds=None):
        rs = self.rs
        r1 = rs.rand(P, n)
        donors = np.empty((k, P), dtype=np.int64)
        irand = np.empty(P, dtype=np.int64)
        resample = np.empty((P, n)) if resample_bounds is not None else None
        for i in range(P):
            p = rs.permutation(P - 1)[:k]
            donors[:, i] = p + (p >= i)
            irand[i] = rs.randint(n)
            if resample is not None:
                resample[i] = rs.uniform(resample_bounds[0], resample_bounds[1], n)
        return {"r1": r1, "donors": donors, "irand": irand, "resample": resample}

    # -- PSO generation: cpso/_cpso.py:262-263 ---------------------------------
    def pso_generation(self, gen, P, n, row0=0):
        r1 = self.rs.rand(P, n)
        r2 = self.rs.rand(P, n)
        return r1, r2

    # -- CPSO restart: cpso/_cpso.py:422, rows in descending-fitness order -----
    def restart_rows(self, gen, lower, upper, rows, n, row0=0):
        return self.rs.uniform(lower, upper, (len(rows), n))

    # -- CMA-ES: cmaes/_cmaes.py:180 and :232-237 ------------------------------
    def cma_initial_mean(self, n):
        return self.rs.uniform(-1.0, 1.0, n)

    def cma_normals(self, gen, P, n, row0=0):
        return np.array([self.rs.randn(n) for _ in range(P)])

    def vd_initial_direction(self, n):
        """vdcma/_vdcma.py:208: np.random.normal(0, 1, n) right after the initial mean."""
        return self.rs.normal(0.0, 1.0, n)

    # -- NA: na/_na.py:298 np.random.uniform(low, high), one per (individual, free axis), individual-major --------
    def na_uniforms(self, gen, P, n, free):
        """The [0, 1) doubles behind the generation's uniform(low, high) = low + (high - low) * u calls, as a (P, n)
        array (columns of fixed axes are never drawn and stay 0)."""
        u = np.zeros((P, n))
        u[:, free] = self.rs.random_sample((P, int(np.count_nonzero(free))))
        return u

    def vd_injection_normals(self, gen, P, n):
        """vdcma/_vdcma.py:245: one more randn(n) per generation once injection is on, after the P x n block."""
        return self.rs.randn(n)


class PhiloxStream:
    """Counter-based draws, identical to the HIP kernels' device generator.

    key = (seed & 0xffffffff, seed >> 32); counter = (slot, row, gen, purpose).
    A row is owned by LPR = 16 / 32 / 64 lanes (n <= 64 / <= 128 / larger); element e belongs to
    lane l = e % LPR at step q = e // LPR.
    Every uniform is 53 bits wide, like numpy's legacy doubles: one call yields d0 = u53(w0, w1), d1 = u53(w2, w3);
    slot = (q >> 1) * LPR + l, half = q & 1 -- since round 6 also the DE crossover uniforms (purpose 0) and PSO's r1 / r2
    (purposes 3 / 4), which rounds 1-5 drew 32 bits wide (profiles/r6_philox53.txt).
    Initial population: lhs_population (counter-based Latin hypercube, every row on its own); the CMA-ES / VD-CMA
    initial mean (n numbers) comes from a private legacy stream.
    """

    kind = "philox"

    def __init__(self, seed):
        if seed is None:
            raise ValueError("philox draws need an explicit seed")
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.k0 = self.seed & 0xFFFFFFFF
        self.k1 = self.seed >> 32
        self.init = LegacyStream(int(seed) & 0xFFFFFFFF)

    def lhs_draws(self, P, n):
        return self.init.lhs_draws(P, n)

    def lhs_population(self, P, n, lower, upper, row0=0, rows=None):
        """The initial population of the throughput mode (csrc/sx_core.hip philox_lhs_kernel): the reference's Latin
        hypercube (_common.py:109-120: strata of width 2/P, jitter of width 1/P, the same scaling arithmetic) with
        counter-based draws, so that rows [row0, row0 + rows) can be produced on their own (a rank draws only its
        shard).  Row i of column j takes stratum sigma_j(i): three rounds of x -> (x*m + a) mod 2^b, x ^= x >> ceil(b/2)
        on b = bit_length(P - 1) bits (m odd; keys = words of two Philox calls with slot j), cycle-walked into
        [0, P); the jitter is a 53-bit uniform keyed by (global row, element)."""
        rows = P - row0 if rows is None else rows
        gi = np.arange(row0, row0 + rows, dtype=np.uint64)
        cols = np.arange(n, dtype=np.uint64)
        ka = philox4x32_10(cols, 0, 0, PURPOSE_INIT_PERM, self.k0, self.k1)
        kb = philox4x32_10(cols, 1, 0, PURPOSE_INIT_PERM, self.k0, self.k1)
        m = [(ka[k] | np.uint64(1))[None, :] for k in range(3)]
        ad = [kb[k][None, :] for k in range(3)]
        b = max(1, int(P - 1).bit_length())
        mask = np.uint64((1 << b) - 1)
        sh = np.uint64((b + 1) // 2)
        x = np.broadcast_to(gi[:, None], (rows, n)).copy()
        todo = np.ones((rows, n), dtype=bool)
        while todo.any():
            y = x
            for k in range(3):
                y = (y * m[k] + ad[k]) & mask  # (below 2^63: x < 2^31, m < 2^32)
                y = y ^ (y >> sh)
            x = np.where(todo, y, x)
            todo &= x >= np.uint64(P)
        u = self._uniform_block(gi, n, 0, PURPOSE_INIT_JITTER)
        step = 2.0 / P  # np.linspace(-1, 1, P, endpoint=False) = arange(P) * step + (-1)
        v = u / P + (x.astype(np.float64) * step + -1.0)
        pop = v * (0.5 * (upper - lower))
        pop += 0.5 * (upper + lower)
        return pop

    def cma_initial_mean(self, n):
        return self.init.cma_initial_mean(n)

    def vd_initial_direction(self, n):
        return self.init.vd_initial_direction(n)

    def vd_injection_normals(self, gen, P, n):
        """The injection draw is "row P" of the generation's normals (one row past the population)."""
        return self.cma_normals(gen, 1, n, row0=P)[0]

    @staticmethod
    def lanes_per_row(n):
        """LPR lanes of a wavefront own one row (csrc/sx_device.hpp lanes_per_row)."""
        return 16 if n <= 64 else (32 if n <= 128 else 64)

    @classmethod
    def _lanes(cls, n):
        """Element e sits in lane l = e % LPR of its row at step q = e // LPR."""
        lpr = np.uint64(cls.lanes_per_row(n))
        e = np.arange(n, dtype=np.uint64)[None, :]
        return e // lpr, e % lpr

    def _uniform_block(self, rows, n, gen, purpose):
        """53-bit uniforms: slot = (q >> 1) * 64 + l, half = q & 1."""
        rows = np.asarray(rows, dtype=np.uint64)[:, None]
        q, l = self._lanes(n)
        slot = (q >> np.uint64(1)) * np.uint64(self.lanes_per_row(n)) + l
        half = (q & np.uint64(1)).astype(bool)
        w0, w1, w2, w3 = philox4x32_10(slot, rows, gen, purpose, self.k0, self.k1)
        return np.where(half, u53(w2, w3), u53(w0, w1))

    def de_generation(self, gen, P, n, k, resample_bounds=None, row0=0):
        rs = self.rs
        r1 = rs.rand(P, n)
        donors = np.empty((k, P), dtype=np.int64)
        for i in range(P):
            # permutation(delete(arange(P), i)) == delete(...)[permutation(P-1)]: same draws
            p = rs.permutation(P - 1)[:k]
            donors[:, i] = p + (p >= i)
        irand = rs.randint(n, size=P)
        resample = None
        if resample_bounds is not None:
            lo, hi = resample_bounds
            resample = rs.uniform(lo, hi, (P, n))
        return {"r1": r1, "donors": donors, "irand": irand, "resample": resample}

    # -- DE generation, updating="immediate": de/_de.py:250, then PER INDIVIDUAL :376 (donor permutation),
    #    :380 (randint(ndim)) and the constraint's uniform(lower, upper, (n,)) (de/_constraints.py:24) --------
    def de_generation_async(self, gen, P, n, k, resample_bounds=None):
        rs = self.rs
        r1 = rs.rand(P, n)
        donors = np.empty((k, P), dtype=np.int64)
        irand = np.empty(P, dtype=np.int64)
        resample = np.empty((P, n)) if resample_bounds is not None else None
        for i in range(P):
            p = rs.permutation(P - 1)[:k]
            donors[:, i] = p + (p >= i)
            irand[i] = rs.randint(n)
            if resample is not None:
                resample[i] = rs.uniform(resample_bounds[0], resample_bounds[1], n)
        return {"r1": r1, "donors": donors, "irand": irand, "resample": resample}

    # -- PSO generation: cpso/_cpso.py:262-263 ---------------------------------
    def pso_generation(self, gen, P, n, row0=0):
        r1 = self.rs.rand(P, n)
        r2 = self.rs.rand(P, n)
        return r1, r2

    # -- CPSO restart: cpso/_cpso.py:422, rows in descending-fitness order -----
    def restart_rows(self, gen, lower, upper, rows, n, row0=0):
        return self.rs.uniform(lower, upper, (len(rows), n))

    # -- CMA-ES: cmaes/_cmaes.py:180 and :232-237 ------------------------------
    def cma_initial_mean(self, n):
        return self.rs.uniform(-1.0, 1.0, n)

    def cma_normals(self, gen, P, n, row0=0):
        return np.array([self.rs.randn(n) for _ in range(P)])

    def vd_initial_direction(self, n):
        """vdcma/_vdcma.py:208: np.random.normal(0, 1, n) right after the initial mean."""
        return self.rs.normal(0.0, 1.0, n)

    # -- NA: na/_na.py:298 np.random.uniform(low, high), one per (individual, free axis), individual-major --------
    def na_uniforms(self, gen, P, n, free):
        """The [0, 1) doubles behind the generation's uniform(low, high) = low + (high - low) * u calls, as a (P, n)
        array (columns of fixed axes are never drawn and stay 0)."""
        u = np.zeros((P, n))
        u[:, free] = self.rs.random_sample((P, int(np.count_nonzero(free))))
        return u

    def vd_injection_normals(self, gen, P, n):
        """vdcma/_vdcma.py:245: one more randn(n) per generation once injection is on, after the P x n block."""
        return self.rs.randn(n)


class PhiloxStream:
    """Counter-based draws, identical to the HIP kernels' device generator.

    key = (seed & 0xffffffff, seed >> 32); counter = (slot, row, gen, purpose).
    A row is owned by LPR = 16 / 32 / 64 lanes (n <= 64 / <= 128 / larger); element e belongs to
    lane l = e % LPR at step q = e // LPR.
    Every uniform is 53 bits wide, like numpy's legacy doubles: one call yields d0 = u53(w0, w1), d1 = u53(w2, w3);
    slot = (q >> 1) * LPR + l, half = q & 1 -- since round 6 also the DE crossover uniforms (purpose 0) and PSO's r1 / r2
    (purposes 3 / 4), which rounds 1-5 drew 32 bits wide (profiles/r6_philox53.txt).
    Initial population: lhs_population (counter-based Latin hypercube, every row on its own); the CMA-ES / VD-CMA
    initial mean (n numbers) comes from a private legacy stream.
    """

    kind = "philox"

    def __init__(self, seed):
        if seed is None:
            raise ValueError("philox draws need an explicit seed")
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.k0 = self.seed & 0xFFFFFFFF
        self.k1 = self.seed >> 32
        self.init = LegacyStream(int(seed) & 0xFFFFFFFF)

    def lhs_draws(self, P, n):
        return self.init.lhs_draws(P, n)

    def lhs_population(self, P, n, lower, upper, row0=0, rows=None):
        """The initial population of the throughput mode (csrc/sx_core.hip philox_lhs_kernel): the reference's Latin
        hypercube (_common.py:109-120: strata of width 2/P, jitter of width 1/P, the same scaling arithmetic) with
        counter-based draws, so that rows [row0, row0 + rows) can be produced on their own (a rank draws only its
        shard).  Row i of column j takes stratum sigma_j(i): three rounds of x -> (x*m + a) mod 2^b, x ^= x >> ceil(b/2)
        on b = bit_length(P - 1) bits (m odd; keys = words of two Philox calls with slot j), cycle-walked into
        [0, P); the jitter is a 53-bit uniform keyed by (global row, element)."""
        rows = P - row0 if rows is None else rows
        gi = np.arange(row0, row0 + rows, dtype=np.uint64)
        cols = np.arange(n, dtype=np.uint64)
        ka = philox4x32_10(cols, 0, 0, PURPOSE_INIT_PERM, self.k0, self.k1)
        kb = philox4x32_10(cols, 1, 0, PURPOSE_INIT_PERM, self.k0, self.k1)
        m = [(ka[k] | np.uint64(1))[None, :] for k in range(3)]
        ad = [kb[k][None, :] for k in range(3)]
        b = max(1, int(P - 1).bit_length())
        mask = np.uint64((1 << b) - 1)
        sh = np.uint64((b + 1) // 2)
        x = np.broadcast_to(gi[:, None], (rows, n)).copy()
        todo = np.ones((rows, n), dtype=bool)
        while todo.any():
            y = x
            for k in range(3):
                y = (y * m[k] + ad[k]) & mask  # (below 2^63: x < 2^31, m < 2^32)
                y = y ^ (y >> sh)
            x = np.where(todo, y, x)
            todo &= x >= np.uint64(P)
        u = self._uniform_block(gi, n, 0, PURPOSE_INIT_JITTER)
        step = 2.0 / P  # np.linspace(-1, 1, P, endpoint=False) = arange(P) * step + (-1)
        v = u / P + (x.astype(np.float64) * step + -1.0)
        pop = v * (0.5 * (upper - lower))
        pop += 0.5 * (upper + lower)
        return pop

    def cma_initial_mean(self, n):
        return self.init.cma_initial_mean(n)

    def vd_initial_direction(self, n):
        return self.init.vd_initial_direction(n)

    def vd_injection_normals(self, gen, P, n):
        """The injection draw is "row P" of the generation's normals (one row past the population)."""
        return self.cma_normals(gen, 1, n, row0=P)[0]

    @staticmethod
    def lanes_per_row(n):
        """LPR lanes of a wavefront own one row (csrc/sx_device.hpp lanes_per_row)."""
        return 16 if n <= 64 else (32 if n <= 128 else 64)

    @classmethod
    def _lanes(cls, n):
        """Element e sits in lane l = e % LPR of its row at step q = e // LPR."""
        lpr = np.uint64(cls.lanes_per_row(n))
        e = np.arange(n, dtype=np.uint64)[None, :]
        return e // lpr, e % lpr

    def _uniform_block(self, rows, n, gen, purpose):
        """53-bit uniforms: slot = (q >> 1) * 64 + l, half = q & 1."""
        rows = np.asarray(rows, dtype=np.uint64)[:, None]
        q, l = self._lanes(n)
        slot = (q >> np.uint64(1)) * np.uint64(self.lanes_per_row(n)) + l
        half = (q & np.uint64(1)).astype(bool)
        w0, w1, w2, w3 = philox4x32_10(slot, rows, gen, purpose, self.k0, self.k1)
        return np.where(half, u53(w2, w3), u53(w0, w1))

    def _uniform32_block(self, rows, n, gen, purpose):
        """32-bit uniforms word * 2^-32: slot = (q >> 2) * 64 + l, word = q & 3."""
        rows = np.asarray(rows, dtype=np.uint64)[:, None]
        q, l = self._lanes(n)
        slot = (q >> np.uint64(2)) * np.uint64(self.lanes_per_row(n)) + l
        wi = (q & np.uint64(3)).astype(np.int64)
        w = philox4x32_10(slot, rows, gen, purpose, self.k0, self.k1)
        pick = np.choose(np.broadcast_to(wi, w[0].shape), w)
        return pick.astype(np.float64) / 4294967296.0

    def de_generation(self, gen, P, n, k, resample_bounds=None, row0=0):
        rows = np.arange(P, dtype=np.uint64) + np.uint64(row0)
        r1 = self._uniform_block(rows, n, gen, PURPOSE_DE_CROSS)  # rand(P, n), de/_de.py:250
        a = philox4x32_10(0, rows, gen, PURPOSE_DE_DONOR, self.k0, self.k1)
        b = philox4x32_10(1, rows, gen, PURPOSE_DE_DONOR, self.k0, self.k1)
        words = list(a[1:]) + list(b)  # word 0 -> irand, word 1+t -> donor t
        donors = np.empty((k, P), dtype=np.int64)
        local = np.arange(P, dtype=np.int64)
        excl = local[None, :].copy()  # sorted exclusion list per row, grows by one per donor
        for t in range(k):
            v = _mulhi(words[t], P - 1 - t)
            for s in range(excl.shape[0]):
                v = v + (v >= excl[s])
            donors[t] = v
            excl = np.sort(np.vstack([excl, v[None, :]]), axis=0)
        irand = _mulhi(a[0], n)
        resample = None
        if resample_bounds is not None:
            lo, hi = resample_bounds
            d = self._uniform_block(rows, n, gen, PURPOSE_DE_RESAMPLE)
            resample = lo + (hi - lo) * d
        return {"r1": r1, "donors": donors, "irand": irand, "resample": resample}

    def de_generation_async(self, gen, P, n, k, resample_bounds=None):
        """Counter-based draws do not depend on the order they are consumed in: the synchronous block."""
        return self.de_generation(gen, P, n, k, resample_bounds)

    def pso_generation(self, gen, P, n, row0=0):
        """r1 and r2 (cpso/_cpso.py:262-263): two blocks of 53-bit uniforms, purposes 3 and 4."""
        rows = np.arange(P, dtype=np.uint64) + np.uint64(row0)
        return (self._uniform_block(rows, n, gen, PURPOSE_PSO_R1), self._uniform_block(rows, n, gen, PURPOSE_PSO_R2))

    def restart_rows(self, gen, lower, upper, rows, n, row0=0):
        d = self._uniform_block(np.asarray(rows, dtype=np.uint64) + np.uint64(row0), n, gen, PURPOSE_PSO_RESTART)
        return lower + (upper - lower) * d

    def na_uniforms(self, gen, P, n, free):
        """53-bit uniforms keyed by (row, generation): element (i, j) of the block layout of _uniform_block."""
        return self._uniform_block(np.arange(P, dtype=np.uint64), n, gen, PURPOSE_NA_UNIFORM)

    def cma_normals(self, gen, P, n, row0=0):
        """Box-Muller on the two doubles of a call: half 0 -> cos branch, half 1 -> sin branch."""
        rows = (np.arange(P, dtype=np.uint64) + np.uint64(row0))[:, None]
        q, l = self._lanes(n)
        slot = (q >> np.uint64(1)) * np.uint64(self.lanes_per_row(n)) + l
        half = (q & np.uint64(1)).astype(bool)
        w0, w1, w2, w3 = philox4x32_10(slot, rows, gen, PURPOSE_CMA_NORMAL, self.k0, self.k1)
        d0 = u53(w0, w1)
        d1 = u53(w2, w3)
        rad = np.sqrt(-2.0 * np.log(1.0 - d0))
        ang = 6.283185307179586 * d1
        return np.where(half, rad * np.sin(ang), rad * np.cos(ang))
