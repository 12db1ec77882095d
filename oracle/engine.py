"""Generation loops of the hot path, restated in numpy (oracle; test infrastructure).

Reference functions followed (paths relative to the reference checkout):
    stochopy/optimize/_common.py:109-120   lhs
    stochopy/optimize/_common.py:123-160   selection_sync
    stochopy/optimize/de/_de.py:176-301    de loop, :314-351 de_sync
    stochopy/optimize/de/_strategy.py:1-38 rand1bin/rand2bin/best1bin/best2bin
    stochopy/optimize/de/_constraints.py:13-28  Random
    stochopy/optimize/cpso/_cpso.py:182-321 cpso loop, :324-329 mutation,
                                     :332-361 pso_sync, :405-426 restart
    stochopy/optimize/cpso/_constraints.py:4-10, 44-53  NoConstraint / Shrink (sync form)
    stochopy/optimize/cmaes/_cmaes.py:143-357 cmaes loop, :360-434 converge
Only synchronous ("deferred") updating is restated: it is what every parallel
backend of the reference runs (de/_de.py:142-145, cpso/_cpso.py:147-150).

The loops take the random draws from a stream object (oracle/streams.py) so the
same arithmetic serves the numpy-legacy stream (reference parity) and the
Philox layout (CPU<->GPU parity at sizes the reference cannot run).
"""
import numpy as np

from .objectives import OBJECTIVES
from .streams import LegacyStream, PhiloxStream

MESSAGES = {
    -8: "TolX",
    -7: "TolFun",
    -6: "TolXUp",
    -5: "EqualFunValues",
    -4: "ConditionCov",
    -3: "NoEffectCoord",
    -2: "NoEffectAxis",
    -1: "maximum number of iterations is reached",
    0: "best solution changes less than xtol",
    1: "best solution value is lower than ftol",
}

DONORS = {"rand1bin": 3, "rand2bin": 5, "best1bin": 2, "best2bin": 4}


class Result(dict):
    __getattr__ = dict.__getitem__


def make_stream(rng, seed):
    if rng in (None, "numpy-legacy"):
        return LegacyStream(seed)
    if rng == "philox":
        return PhiloxStream(seed)
    raise ValueError(rng)


def latin_hypercube(stream, P, n, lower, upper):
    """_common.py:109-120 (jitter is 1/P wide inside strata 2/P wide -- reproduced as is).  Philox mode: the same
    construction with counter-based draws (PhiloxStream.lhs_population), as the HIP path's philox_lhs_kernel."""
    if hasattr(stream, "lhs_population"):
        return stream.lhs_population(P, n, np.asarray(lower, dtype=np.float64), np.asarray(upper, dtype=np.float64))
    u, perms = stream.lhs_draws(P, n)
    x = u / P
    x += np.linspace(-1.0, 1.0, P, endpoint=False)[:, None]
    pop = np.empty((P, n))
    for j in range(n):
        pop[:, j] = x[perms[j], j]
    pop *= 0.5 * (upper - lower)
    pop += 0.5 * (upper + lower)
    return pop


def termination(it, maxiter, dx, fbest, xtol, ftol):
    """Status ladder of _common.py:134-158: 0, then 1, then -1, else None."""
    if dx <= xtol and fbest <= ftol:
        return 0
    if fbest <= ftol:
        return 1
    if it >= maxiter:
        return -1
    return None


def greedy_select(it, cand, candfun, xbest, x, xfun, maxiter, xtol, ftol):
    """_common.py:127-158 after the evaluation: strict <, in-place, first argmin."""
    better = candfun < xfun
    xfun[better] = candfun[better]
    x[better] = cand[better]
    k = int(np.argmin(xfun))
    dx = np.linalg.norm(xbest - x[k])
    status = termination(it, maxiter, dx, xfun[k], xtol, ftol)
    return x[k].copy(), xfun[k], status


def async_select(u, fu, i, x, xfun, xbest, xbestfun, xtol, ftol):
    """_common.py:163-194 for individual i after its evaluation: `<=` acceptance, and the best row / status
    are updated on the spot.  Returns (xbest, xbestfun, status of THIS individual)."""
    status = None
    if fu <= xfun[i]:
        x[i] = u
        xfun[i] = fu
        if fu <= xbestfun:
            if fu <= ftol:
                status = 0 if np.linalg.norm(xbest - u) <= xtol else 1
            xbest = u.copy()
            xbestfun = fu
    return xbest, xbestfun, status


class History:
    """return_all bookkeeping of de/_de.py:221-234, 270-278 (same for cpso)."""

    def __init__(self, enabled, maxiter, P, n, verbosity):
        self.enabled = enabled
        if enabled:
            self.nout = int(np.ceil(verbosity * P))
            rows = max(self.nout, 1)
            self.xall = np.empty((maxiter, rows, n))
            self.funall = np.empty((maxiter, rows))

    def put(self, slot, X, f, gbest=None, gfit=None):
        if not self.enabled:
            return
        if self.nout > 0:
            self.xall[slot] = X[: self.nout]
            self.funall[slot] = f[: self.nout]
        elif gbest is not None:
            self.xall[slot] = gbest
            self.funall[slot] = gfit
        else:
            k = int(np.argmin(f))
            self.xall[slot] = X[k]
            self.funall[slot] = f[k]

    def fill(self, res, it):
        if self.enabled:
            res["xall"] = self.xall[:it]
            res["funall"] = self.funall[:it]


def _final(x, fun, status, nfev, nit, hist):
    res = Result(x=x, success=status >= 0, status=status, message=MESSAGES[status], fun=fun, nfev=nfev, nit=nit)
    hist.fill(res, nit)
    return res


# --------------------------------------------------------------------------- #
# DE  (de/_de.py:176-301, 314-351)
# --------------------------------------------------------------------------- #
def de_mutants(strategy, d, F, X, gbest):
    """de/_strategy.py:1-38, same association order."""
    if strategy == "rand1bin":
        return X[d[0]] + F * (X[d[1]] - X[d[2]])
    if strategy == "rand2bin":
        return X[d[0]] + F * (X[d[1]] + X[d[2]] - X[d[3]] - X[d[4]])
    if strategy == "best1bin":
        return gbest + F * (X[d[0]] - X[d[1]])
    if strategy == "best2bin":
        return gbest + F * (X[d[0]] + X[d[1]] - X[d[2]] - X[d[3]])
    raise KeyError(strategy)


def de_candidates(X, gbest, draws, F, CR, strategy, lower, upper, constraints):
    """de/_de.py:333-344: mutation, binomial crossover (r1 <= CR, forced index), bound repair."""
    P = X.shape[0]
    V = de_mutants(strategy, draws["donors"], F, X, gbest)
    take = draws["r1"] <= CR
    take[np.arange(P), draws["irand"]] = True
    U = np.where(take, V, X)
    if constraints == "Random":
        U = np.where((U < lower) | (U > upper), draws["resample"], U)  # de/_constraints.py:21-26
    return U


def run_de(fobj, lower, upper, x0, stream, callback=None, maxiter=100, popsize=10, mutation=0.5,
           recombination=0.9, strategy="best1bin", xtol=1e-8, ftol=1e-8, constraints=None,
           return_all=False, verbosity=1.0, updating="deferred", **_ignored):
    n = len(lower)
    P = popsize
    k = DONORS[strategy]
    immediate = updating == "immediate"
    X = np.array(x0, dtype=np.float64) if x0 is not None else latin_hypercube(stream, P, n, lower, upper)
    pfit = fobj(X)
    fit = pfit.copy()
    g = int(np.argmin(fit))
    gfit = fit[g]
    gbest = X[g].copy()
    hist = History(return_all, maxiter, P, n, verbosity)
    hist.put(0, X, pfit, gbest, gfit)
    if callback is not None:
        callback(X, Result(x=gbest, fun=gfit, nfev=P, nit=1))
    it = 1
    while True:
        it += 1
        rb = (lower, upper) if constraints == "Random" else None
        if immediate:
            # de/_de.py:354-391: one individual at a time, on the population as it stands
            draws = stream.de_generation_async(it, P, n, k, rb)
            pfit = np.empty(P)
            for i in range(P):
                V = de_mutants(strategy, draws["donors"][:, i], mutation, X, gbest)
                take = draws["r1"][i] <= recombination
                take[draws["irand"][i]] = True
                u = np.where(take, V, X[i])
                if constraints == "Random":
                    u = np.where((u < lower) | (u > upper), draws["resample"][i], u)
                pfit[i] = fobj(u[None, :])[0]
                gbest, gfit, status = async_select(u, pfit[i], i, X, fit, gbest, gfit, xtol, ftol)
            if status is None and it >= maxiter:
                status = -1
        else:
            draws = stream.de_generation(it, P, n, k, rb)
            U = de_candidates(X, gbest, draws, mutation, recombination, strategy, lower, upper, constraints)
            pfit = fobj(U)  # NB de/_de.py:270-273: funall pairs post-selection X with CANDIDATE fitness
            gbest, gfit, status = greedy_select(it, U, pfit, gbest, X, fit, maxiter, xtol, ftol)
        hist.put(it - 1, X, pfit)
        if callback is not None:
            callback(X, Result(x=gbest, fun=gfit, nfev=it * P, nit=it))
        if status is not None:
            break
    return _final(gbest, gfit, status, it * P, it, hist)


# --------------------------------------------------------------------------- #
# PSO / CPSO  (cpso/_cpso.py:182-321, 324-361, 405-426)
# --------------------------------------------------------------------------- #
def shrink_factor(X, V, lower, upper):
    """cpso/_constraints.py:22-53 (sync form): per-row min over violated dims, else 1."""
    Xc = X + V
    with np.errstate(divide="ignore", invalid="ignore"):
        bl = np.where(Xc < lower, (lower - X) / V, np.inf)
        bu = np.where(Xc > upper, (upper - X) / V, np.inf)
    beta = np.minimum(bl.min(axis=1), bu.min(axis=1))
    return np.where(np.isinf(beta), 1.0, beta)


def pso_move(X, V, pbest, gbest, w, c1, c2, r1, r2, lower, upper, constraints):
    """cpso/_cpso.py:324-329, left-to-right association; then the constraint."""
    V = w * V + c1 * r1 * (pbest - X) + c2 * r2 * (gbest - X)
    if constraints == "Shrink":
        V = V * shrink_factor(X, V, lower, upper)[:, None]
    return X + V, V


def swarm_radius(X, gbest, n):
    """cpso/_cpso.py:410-411."""
    d = X - gbest
    return np.sqrt((d * d).sum(axis=1)).max() / np.sqrt(4.0 * n)


def restart_count(it, maxiter, P, gamma):
    """cpso/_cpso.py:415-416."""
    inorm = it / maxiter
    return int((P - 1.0) / (1.0 + np.exp(1.0 / 0.09 * (inorm - gamma + 0.5))))


def run_pso(fobj, lower, upper, x0, stream, callback=None, maxiter=100, popsize=10, inertia=0.7298,
            cognitivity=1.49618, sociability=1.49618, competitivity=None, xtol=1e-8, ftol=1e-8,
            constraints=None, return_all=False, verbosity=1.0, updating="deferred", **_ignored):
    n = len(lower)
    P = popsize
    gamma = competitivity
    immediate = updating == "immediate"
    if gamma:
        delta = np.log(1.0 + 0.003 * P) / np.max((0.2, np.log(0.01 * maxiter)))
    X = np.array(x0, dtype=np.float64) if x0 is not None else latin_hypercube(stream, P, n, lower, upper)
    V = np.zeros((P, n))
    pbest = X.copy()
    pfit = fobj(X)
    pbestfit = pfit.copy()
    g = int(np.argmin(pbestfit))
    gfit = pbestfit[g]
    gbest = X[g].copy()
    hist = History(return_all, maxiter, P, n, verbosity)
    hist.put(0, X, pfit, gbest, gfit)
    if callback is not None:
        callback(X, Result(x=gbest, fun=gfit, nfev=P, nit=1))
    it = 1
    restarts, restart_rows = [], []
    while True:
        it += 1
        r1, r2 = stream.pso_generation(it, P, n)
        if immediate:
            # cpso/_cpso.py:364-402: particle by particle, each against the best row as it stands
            pfit = np.empty(P)
            for i in range(P):
                xi, vi = pso_move(X[i : i + 1], V[i : i + 1], pbest[i : i + 1], gbest, inertia, cognitivity,
                                  sociability, r1[i : i + 1], r2[i : i + 1], lower, upper, constraints)
                X[i], V[i] = xi[0], vi[0]
                pfit[i] = fobj(X[i : i + 1])[0]
                gbest, gfit, status = async_select(X[i].copy(), pfit[i], i, pbest, pbestfit, gbest, gfit, xtol, ftol)
            if status is None and it >= maxiter:
                status = -1
        else:
            X, V = pso_move(X, V, pbest, gbest, inertia, cognitivity, sociability, r1, r2, lower, upper, constraints)
            pfit = fobj(X)
            gbest, gfit, status = greedy_select(it, X, pfit, gbest, pbest, pbestfit, maxiter, xtol, ftol)
        hist.put(it - 1, X, pfit)
        if callback is not None:
            callback(X, Result(x=gbest, fun=gfit, nfev=it * P, nit=it))
        if status is not None:
            break
        if gamma:
            # d = X - gbest; per-row sqrt(dot) exactly as np.linalg.norm does for 1-D input
            rad = (max(np.linalg.norm(X[i] - gbest) for i in range(P)) / np.sqrt(4.0 * n)) if P <= 32768 else swarm_radius(X, gbest, n)
            if rad < delta:
                nw = restart_count(it, maxiter, P, gamma)
                if nw > 0:
                    rows = pbestfit.argsort()[: -nw - 1 : -1]
                    V[rows] = 0.0
                    X[rows] = stream.restart_rows(it, lower, upper, rows, n)
                    pbest[rows] = X[rows]
                    pbestfit[rows] = 1.0e30
                    restarts.append((it, nw))
                    restart_rows.append(np.sort(rows))
    res = _final(gbest, gfit, status, it * P, it, hist)
    res["_restarts"] = restarts
    res["_restart_rows"] = restart_rows  # which rows each restart re-seeded (sorted): pinned by tests/golden/configs_long
    return res


# --------------------------------------------------------------------------- #
# CMA-ES  (cmaes/_cmaes.py:143-357, 360-434)
# --------------------------------------------------------------------------- #
def cma_constants(n, P, muperc):
    """cmaes/_cmaes.py:184-205."""
    mu = int(muperc * P)
    w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    w /= w.sum()
    mueff = w.sum() ** 2 / np.square(w).sum()
    cc = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)
    cs = (mueff + 2.0) / (n + mueff + 5.0)
    c1 = 2.0 / ((n + 1.3) ** 2 + mueff)
    cmu = min(1.0 - c1, 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))
    damps = 1.0 + 2.0 * max(0.0, np.sqrt((mueff - 1.0) / (n + 1.0)) - 1.0) + cs
    chind = np.sqrt(n) * (1.0 - 1.0 / (4.0 * n) + 1.0 / (21.0 * n**2))
    return dict(mu=mu, w=w, mueff=mueff, cc=cc, cs=cs, c1=c1, cmu=cmu, damps=damps, chind=chind)


def cma_stop(it, n, maxiter, xmean, xold, bestfit_hist, arfit, order, sigma, insigma, ilim, pc, xtol, ftol, diagC, B, D):
    """cmaes/_cmaes.py:360-434, the ten ordered rules (incl. the zero-padding artefacts)."""
    i = int(np.floor(np.mod(it, n)))
    sq = np.sqrt(diagC)
    fb = arfit[order[0]]
    if it >= maxiter:
        return -1
    if np.linalg.norm(xold - xmean) <= xtol and fb < ftol:
        return 0
    if fb <= ftol:
        return 1
    if B is not None and (np.abs(0.1 * sigma * B[:, i] * D[i]) < 1.0e-10).all():
        return -2
    if (0.2 * sigma * sq < 1.0e-10).any():
        return -3
    if D is not None and D.max() > 1.0e7 * D.min():
        return -4
    if it >= ilim:
        win = bestfit_hist[it - ilim : it + 1]
        if win.max() - win.min() < 1.0e-10:
            return -5
    if (sigma * sq > 1.0e3 * insigma).any():
        return -6
    if it > 2:
        both = np.append(arfit, bestfit_hist)
        if both.max() - both.min() < 1.0e-12:
            return -7
    if (sigma * np.append(np.abs(pc), sq.max()) < 1.0e-11 * insigma).all():
        return -8
    return None


def cma_sample(xmean, sigma, B, D, Z):
    """cmaes/_cmaes.py:232-237, row by row as the reference does (matvec per individual)."""
    return np.array([xmean + sigma * np.dot(B, D * z) for z in Z])


def cma_covariance(C, arx_sel, xold, sigma, w, pc, cond, c1, cmu, cc):
    """cmaes/_cmaes.py:290-295 (tmp uses the PRE-update C)."""
    mu = arx_sel.shape[0]
    artmp = (arx_sel - np.tile(xold, (mu, 1))) / sigma
    tmp = 0.0 if cond else c1 * cc * (2.0 - cc) * C
    C = C * (1.0 - c1 - cmu)
    C += cmu * np.dot(np.dot(artmp.T, np.diag(w)), artmp)
    C += c1 * np.outer(pc, pc)
    C += tmp
    return C


class PenalizeState:
    """CMA-ES box-constraint handling "Penalize" (cmaes/_constraints.py:4-82), restated.  State carried over
    generations (cmaes/_cmaes.py:213-215, 230-231): boundary weights, the history of fitness-spread estimates,
    and the two phase flags."""

    def __init__(self, n):
        self.weights = np.zeros(n)
        self.dfithist = np.ones(1)
        self.validfitval = False
        self.iniphase = True

    def apply(self, arx, xmean, xold, sigma, diagC, mueff, it, fun):
        P, n = arx.shape
        valid = np.clip(arx, -1.0, 1.0)                      # :29-31
        fit = fun(valid)
        q25, q75 = np.percentile(fit, [25.0, 75.0])          # :34-35
        delta = (q75 - q25) / n / diagC.mean() / sigma**2
        if delta == 0:                                       # :38-42
            delta = self.dfithist[self.dfithist > 0.0].min()
        elif not self.validfitval:
            self.dfithist = np.empty(0)
            self.validfitval = True
        if self.dfithist.size < 20 + (3.0 * n) / P:          # :45-48 (sliding window)
            self.dfithist = np.append(self.dfithist, delta)
        else:
            self.dfithist = np.append(self.dfithist[1:], delta)
        outside = (xmean < -1.0) | (xmean > 1.0)             # :51
        # :52-53 -- the second assignment overwrites the first, so only the UPPER side is clipped here
        tx = np.where(xmean > 1.0, 1.0, xmean)
        if self.iniphase and outside.any():                  # :56-59
            self.weights = np.full(n, 2.0002 * np.median(self.dfithist))
            if self.validfitval and it > 2:
                self.iniphase = False
        if outside.any():                                    # :61-73
            tx = xmean - tx
            grow = outside & (np.abs(tx) > 3.0 * max(1.0, np.sqrt(n / mueff)) * sigma * np.sqrt(diagC))
            grow &= np.sign(tx) == np.sign(xmean - xold)
            self.weights = np.where(grow, self.weights * 1.2 ** min(1.0, mueff / 10.0 / n), self.weights)
        scale = np.exp(0.9 * (np.log(diagC) - np.log(diagC).mean()))   # :76
        fit = fit + np.dot((valid - arx) ** 2, self.weights / scale)   # :79
        return fit, valid


# --------------------------------------------------------------------------- #
# NA  (na/_na.py:131-262 loop, :265-305 mutation)
# --------------------------------------------------------------------------- #
def na_walk(models, k, u_row, free):
    """One new sample (na/_na.py:275-303): start at model k, then axis by axis draw uniformly inside the Voronoi
    cell of k restricted to the current axis line.  `u_row[j]` is the [0,1) double behind np.random.uniform(low,
    high) (:298).  The cell-distance bookkeeping `d1` is SCALAR arithmetic in the reference: numpy scalars `** 2`,
    i.e. libm pow (not always the correctly rounded square) -- kept as numpy scalars here; `d2` is array arithmetic
    (`** 2` on arrays is an exact square)."""
    n = models.shape[1]
    centre = models[k]
    x = centre.copy()
    others = np.delete(models, k, axis=0)
    d1 = 0.0
    d2 = ((others[:, 1:] - x[1:]) ** 2).sum(axis=1)                                  # :280
    for j in range(n):
        if not free[j]:
            x[j] = 0.0  # fixed axis: un-normalisation puts the bound back (:283-286)
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            lim = 0.5 * (centre[j] + others[:, j] + (d1 - d2) / (centre[j] - others[:, j]))   # :288
        below = lim <= x[j]
        low = max(lim[below].max(), 0.0) if below.sum() else 0.0                   # :290-291
        above = lim >= x[j]
        high = min(lim[above].min(), 1.0) if above.sum() else 1.0                  # :293-294
        x[j] = low + (high - low) * u_row[j]                                        # :296 uniform(low, high)
        if j < n - 1:                                                               # :298-303
            d1 += (centre[j] - x[j]) ** 2 - (centre[j + 1] - x[j + 1]) ** 2
            d2 += (others[:, j] - x[j]) ** 2 - (others[:, j + 1] - x[j + 1]) ** 2
    return x


def run_na(fobj, lower, upper, x0, stream, callback=None, maxiter=100, popsize=10, nrperc=0.5, xtol=1e-8, ftol=1e-8,
           return_all=False, verbosity=1.0, **_ignored):
    """Neighbourhood Algorithm, na/_na.py:131-262: all models ever sampled are kept (newest first); every generation
    resamples popsize points inside the Voronoi cells of the nr best of them."""
    n = len(lower)
    P = popsize
    span = upper - lower
    free = span > 0.0
    span = np.where(free, span, 1.0)                                                # :151-153
    normalize = lambda x: np.where(free, (x - lower) / span, upper)  # noqa: E731   :154
    unnormalize = lambda x: np.where(free, x * span + lower, upper)  # noqa: E731   :155
    fun = lambda x: fobj(unnormalize(x))  # noqa: E731
    nr = max(1, int(nrperc * P))                                                    # :160
    X = np.asarray(x0, dtype=np.float64) if x0 is not None else latin_hypercube(stream, P, n, lower, upper)
    X = normalize(X)
    pbest = X.copy()
    pfit = fun(X)
    pbestfit = pfit.copy()
    g = int(np.argmin(pbestfit))
    gfit = pbestfit[g]
    gbest = X[g].copy()
    models, models_fit = X.copy(), pfit.copy()                                      # :177-178
    hist = History(return_all, maxiter, P, n, verbosity)
    hist.put(0, X, pfit, gbest, gfit)  # :185-193 -- NB the reference stores the NORMALISED rows at iteration 1
    if callback is not None:
        callback(unnormalize(X), Result(x=unnormalize(gbest), fun=gfit, nfev=P, nit=1))
    it = 1
    while True:
        it += 1
        u = stream.na_uniforms(it, P, n, free)
        order = models_fit.argsort()[:nr]                                           # :274
        X = np.array([na_walk(models, order[i % nr], u[i], free) for i in range(P)])
        pfit = fun(X)
        gbest, gfit, status = greedy_select(it, X, pfit, gbest, pbest, pbestfit, maxiter, xtol, ftol)
        models = np.vstack((X, models))                                             # :223-224
        models_fit = np.concatenate((pfit, models_fit))
        hist.put(it - 1, unnormalize(X), pfit)
        if callback is not None:
            callback(unnormalize(X), Result(x=unnormalize(gbest), fun=gfit, nfev=it * P, nit=it))
        if status is not None:
            break
    return _final(unnormalize(gbest), gfit, status, it * P, it, hist)


def eigh_canonical(C):
    """cmaes/_cmaes.py:303-305 (upper triangle mirrored, numpy.linalg.eigh, ascending order) followed by the sign
    rule of the device eigensolver (csrc/sx_eigh.hip): the component of largest magnitude of every eigenvector
    (lowest row on ties) is positive.  LAPACK leaves the signs to its internals; a basis that both sides of a
    parity test can reproduce needs a rule, and this is the one the HIP path states.  Eigenvalues are untouched."""
    C = np.triu(C) + np.triu(C, 1).T
    D, B = np.linalg.eigh(C)
    o = np.argsort(D, kind="stable")
    D = D[o]
    B = B[:, o]
    top = np.argmax(np.abs(B), axis=0)  # first occurrence of the maximum = lowest row on ties
    sgn = np.where(B[top, np.arange(B.shape[1])] < 0.0, -1.0, 1.0)
    return D, B * sgn


def run_cmaes(fobj, lower, upper, x0, stream, callback=None, maxiter=100, popsize=10, sigma=0.1, muperc=0.5,
              xtol=1e-8, ftol=1e-8, constraints=None, return_all=False, verbosity=1.0, eigh="lapack", probe=None,
              **_ignored):
    """eigh="lapack" (or a callable wrapping it): the reference's call as is (pinned to the goldens); eigh="canonical": the same
    decomposition with the device eigensolver's sign rule (what eigh="device" runs of the HIP path are compared
    with)."""
    if not callable(eigh) and eigh not in ("lapack", "canonical"):
        raise ValueError(eigh)
    if constraints not in (None, "Penalize"):
        raise KeyError(constraints)
    pen = PenalizeState(len(lower)) if constraints == "Penalize" else None
    n = len(lower)
    P = popsize
    xm = 0.5 * (upper + lower)
    xstd = 0.5 * (upper - lower)
    unstd = lambda x: x * xstd + xm  # noqa: E731  cmaes/_cmaes.py:167-173
    xmean = stream.cma_initial_mean(n) if x0 is None else (np.asarray(x0, dtype=np.float64) - xm) / xstd
    xold = np.empty(n)
    k = cma_constants(n, P, muperc)
    mu, w, mueff, cc, cs, c1, cmu, damps, chind = (k[s] for s in ("mu", "w", "mueff", "cc", "cs", "c1", "cmu", "damps", "chind"))
    pc = np.zeros(n)
    ps = np.zeros(n)
    B = np.eye(n)
    D = np.ones(n)
    C = np.eye(n)
    invsqrtC = np.eye(n)
    hist = History(return_all, maxiter, P, n, verbosity)
    nfev = 0
    eigeneval = 0
    bestfit_hist = np.zeros(maxiter)
    ilim = int(10.0 + 30.0 * n / P)
    insigma = sigma
    it = 0
    while True:
        it += 1
        if probe is not None:  # tests: the model a generation starts from (copies)
            before = dict(xmean=xmean.copy(), sigma=sigma, ps=ps.copy(), pc=pc.copy(), C=C.copy(), B=B.copy(), D=D.copy(),
                          besthist=bestfit_hist.copy())
        Z = stream.cma_normals(it, P, n)
        arx = cma_sample(xmean, sigma, B, D, Z)
        arxvalid = arx
        if pen is None:
            arfit = fobj(unstd(arx))
        else:  # cmaes/_cmaes.py:238-256: the valid (clipped) points are what the caller sees, arx drives the model
            arfit, arxvalid = pen.apply(arx, xmean, xold, sigma, np.diag(C), mueff, it, lambda x: fobj(unstd(x)))
        nfev += P
        hist.put(it - 1, unstd(arxvalid), arfit)
        order = np.argsort(arfit)
        xold = xmean.copy()
        xmean = np.dot(w, arx[order[:mu], :])
        bestfit_hist[it - 1] = arfit[order[0]]
        ps = (1.0 - cs) * ps + np.sqrt(cs * (2.0 - cs) * mueff) * np.dot(invsqrtC, xmean - xold) / sigma
        cond = np.linalg.norm(ps) / np.sqrt(1.0 - (1.0 - cs) ** (2.0 * nfev / P)) / chind < 1.4 + 2.0 / (n + 1.0)
        pc *= 1.0 - cc
        pc += np.sqrt(cc * (2.0 - cc) * mueff) * (xmean - xold) / sigma if cond else 0.0
        C = cma_covariance(C, arx[order[:mu], :], xold, sigma, w, pc, cond, c1, cmu, cc)
        sigma *= np.exp((cs / damps) * (np.linalg.norm(ps) / chind - 1.0))
        due = nfev - eigeneval > P / (c1 + cmu) / n / 10.0
        if due:
            eigeneval = nfev
            C = np.triu(C) + np.triu(C, 1).T
            if eigh == "canonical":
                D, B = eigh_canonical(C)
            else:  # a callable: the same LAPACK call behind a recorder (tests replay the pairs into the HIP run)
                D, B = eigh(C) if callable(eigh) else np.linalg.eigh(C)
                o = np.argsort(D)
                D = D[o]
                B = B[:, o]
            D = np.sqrt(D)
            invsqrtC = np.dot(np.dot(B, np.diag(1.0 / D)), B.T)
        status = cma_stop(it, n, maxiter, xmean, xold, bestfit_hist, arfit, order, sigma, insigma, ilim, pc,
                          xtol, ftol, np.diag(C), B, D)
        if probe is not None:
            probe(it, before, dict(arx=arx.copy(), arfit=arfit.copy(), order=order.copy(), xmean=xmean.copy(), ps=ps.copy(),
                                   pc=pc.copy(), C=C.copy(), sigma=sigma, B=B.copy(), D=D.copy(), due=bool(due),
                                   status=status))
        if callback is not None:
            callback(unstd(arxvalid), Result(x=unstd(arxvalid[order[0]]), fun=arfit[order[0]], nfev=nfev, nit=it))
        if status is not None:
            break
    return _final(unstd(arxvalid[order[0]]), arfit[order[0]], status, nfev, it, hist)


def vd_moments(vn, norm_v2, y, w=None):
    """vdcma/_vdcma.py:428-444: the (weighted) first-order statistics p, q of y under the model D(I + vv^T)D."""
    t = np.dot(y, vn)
    shrink = norm_v2 / (1.0 + norm_v2)
    if w is None:
        p = y**2 - shrink * (t * y * vn) - 1.0
        q = t * y - (0.5 * (t**2 + 1.0 + norm_v2)) * vn
        return p, q
    p = np.dot(w, y**2 - shrink * (t * (y * vn).T).T - 1.0)
    q = np.dot(w, (t * y.T).T - np.outer(0.5 * (t**2 + 1.0 + norm_v2), vn))
    return p, q


def vd_natural_gradient(dvec, vn, vnn, norm_v, norm_v2, alpha, avec, bsca, invavnn, p, q):
    """vdcma/_vdcma.py:447-460: natural-gradient steps for v and d from the moments."""
    r = p - alpha / (1.0 + norm_v2) * ((2.0 + norm_v2) * q * vn - norm_v2 * np.dot(vn, q) * vnn)
    s = r / avec - bsca * np.dot(r, invavnn) / (1.0 + bsca * np.dot(vnn, invavnn)) * invavnn
    ngv = q / norm_v - alpha / norm_v * ((2.0 + norm_v2) * (vn * s) - np.dot(s, vnn) * vn)
    return ngv, dvec * s


def run_vdcma(fobj, lower, upper, x0, stream, callback=None, maxiter=100, popsize=10, sigma=0.1, muperc=0.5,
              xtol=1e-8, ftol=1e-8, constraints=None, return_all=False, verbosity=1.0, probe=None, **_ignored):
    """VD-CMA (vdcma/_vdcma.py:144-425): covariance model D (I + v v^T) D, O(n) per sample."""
    if constraints not in (None, "Penalize"):
        raise KeyError(constraints)
    n = len(lower)
    P = popsize
    pen = PenalizeState(n) if constraints == "Penalize" else None
    xm = 0.5 * (upper + lower)
    xstd = 0.5 * (upper - lower)
    unstd = lambda x: x * xstd + xm  # noqa: E731
    xmean = stream.cma_initial_mean(n) if x0 is None else (np.asarray(x0, dtype=np.float64) - xm) / xstd
    xold = np.empty(n)
    mu = int(muperc * P)
    w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
    w /= w.sum()
    mueff = w.sum() ** 2 / np.square(w).sum()
    cc = (4.0 + mueff / n) / (n + 4.0 + 2.0 * mueff / n)           # :193-199
    cfactor = (n - 5.0) / 6.0
    c1 = cfactor * 2.0 / ((n + 1.3) ** 2 + mueff)
    cmu = min(1.0 - c1, cfactor * 2.0 * (mueff - 2.0 + 1.0 / mueff) / ((n + 2.0) ** 2 + mueff))
    inject = False                                                 # :202-213
    cs, ds = 0.3, np.sqrt(n)
    dx = np.zeros(n)
    ps = 0.0
    dvec = np.ones(n)
    vvec = stream.vd_initial_direction(n) / np.sqrt(n)
    norm_v2 = np.dot(vvec, vvec)
    norm_v = np.sqrt(norm_v2)
    vn = vvec / norm_v
    vnn = vn**2
    pc = np.zeros(n)
    hist = History(return_all, maxiter, P, n, verbosity)
    nfev = 0
    bestfit_hist = np.zeros(maxiter)
    ilim = int(10 + 30 * n / P)
    insigma = sigma
    it = 0
    while True:
        it += 1
        if probe is not None:  # tests: the model a generation starts from (copies)
            before = dict(xmean=xmean.copy(), sigma=sigma, ps=ps, dx=dx.copy(), dvec=dvec.copy(), vvec=vvec.copy(),
                          vn=vn.copy(), pc=pc.copy(), norm_v2=norm_v2, norm_v=norm_v, inject=inject,
                          besthist=bestfit_hist.copy())
        arz = stream.cma_normals(it, P, n)                          # :236-247
        ary = dvec * (arz + (np.sqrt(1.0 + norm_v2) - 1.0) * np.outer(np.dot(arz, vn), vn))
        if inject:
            ddx = dx / dvec
            mnorm = (ddx**2).sum() - np.dot(ddx, vvec) ** 2 / (1.0 + norm_v2)
            dy = np.linalg.norm(stream.vd_injection_normals(it, P, n)) / np.sqrt(mnorm) * dx
            ary[0] = dy
            ary[1] = -dy
        arx = xmean + sigma * ary
        if n <= 2048:
            diagC = np.diag(np.dot(np.dot(np.diag(dvec), np.eye(n) + np.outer(vvec, vvec)), np.diag(dvec)))  # :249-254
        else:
            # The reference forms the dense n x n products above (O(n^3) per generation) just for their diagonal.  Entry i of
            # that diagonal is (d_i * (1 + v_i v_i)) * d_i plus exact zeros (diag(d) has one non-zero per row and column),
            # so the O(n) form is the same number bit for bit -- pinned against the reference's own run at n = 8192
            # (tests/golden/vdcma_wide.json, test_wide_rows_bit_exact); without it a wide test would take minutes per generation.
            diagC = (dvec * (1.0 + vvec * vvec)) * dvec
        arxvalid = arx
        if pen is None:
            arfit = fobj(unstd(arx))
        else:
            arfit, arxvalid = pen.apply(arx, xmean, xold, sigma, diagC, mueff, it, lambda x: fobj(unstd(x)))
        nfev += P
        hist.put(it - 1, unstd(arxvalid), arfit)
        order = np.argsort(arfit)                                   # :289-295
        dx = np.dot(w, arx[order[:mu]]) - w.sum() * xmean
        xold = xmean.copy()
        xmean = xmean + dx
        bestfit_hist[it - 1] = arfit[order[0]]
        if inject:                                                  # :298-306
            rank_gap = np.where(order == 1)[0][0] - np.where(order == 0)[0][0]
            rank_gap = rank_gap / (P - 1.0)
            ps += cs * (rank_gap - ps)
            sigma *= np.exp(ps / ds)
            cond = ps < 0.5
        else:
            inject = True
            cond = True
        pc *= 1.0 - cc                                              # :309-314
        pc += np.sqrt(cc * (2.0 - cc) * mueff) * np.dot(w, ary[order[:mu]]) if cond else 0.0
        gamma = 1.0 / np.sqrt(1.0 + norm_v2)                        # :317-328
        alpha = np.sqrt(norm_v2**2 + (1.0 + norm_v2) / vnn.max() * (2.0 - gamma)) / (2.0 + norm_v2)
        if alpha < 1.0:
            beta = (4.0 - (2.0 - gamma) / vnn.max()) / (1.0 + 2.0 / norm_v2) ** 2
        else:
            alpha, beta = 1.0, 0.0
        bsca = 2.0 * alpha**2 - beta
        avec = 2.0 - (bsca + 2.0 * alpha**2) * vnn
        invavnn = vnn / avec
        if cmu == 0.0:                                              # :331-345
            p_mu, q_mu = np.zeros(n), np.zeros(n)
        else:
            p_mu, q_mu = vd_moments(vn, norm_v2, ary[order[:mu]] / dvec, w)
        if c1 == 0.0:
            p_one, q_one = np.zeros(n), np.zeros(n)
        else:
            p_one, q_one = vd_moments(vn, norm_v2, pc / dvec)
        p = cmu * p_mu                                              # :348-352
        q = cmu * q_mu
        if cond:
            p += c1 * p_one
            q += c1 * q_one
        if cmu + c1 > 0.0:                                          # :355-368
            ngv, ngd = vd_natural_gradient(dvec, vn, vnn, norm_v, norm_v2, alpha, avec, bsca, invavnn, p, q)
            up = min(1.0, 0.7 * norm_v / np.sqrt(np.dot(ngv, ngv)))
            up = min(up, 0.7 * (dvec / np.abs(ngd)).min())
        else:
            ngv, ngd, up = np.zeros(n), np.zeros(n), 1.0
        vvec = vvec + up * ngv                                      # :371-378
        dvec = dvec + up * ngd
        norm_v2 = np.dot(vvec, vvec)
        norm_v = np.sqrt(norm_v2)
        vn = vvec / norm_v
        vnn = vn**2
        status = cma_stop(it, n, maxiter, xmean, xold, bestfit_hist, arfit, order, sigma, insigma, ilim, pc,
                          xtol, ftol, diagC, None, None)
        if probe is not None:
            probe(it, before, dict(arx=arx.copy(), ary=ary.copy(), arfit=arfit.copy(), order=order.copy(), xmean=xmean.copy(),
                                   dx=dx.copy(), sigma=sigma, ps=ps, pc=pc.copy(), dvec=dvec.copy(), vvec=vvec.copy(),
                                   vn=vn.copy(), norm_v2=norm_v2, status=status))
        if callback is not None:
            callback(unstd(arxvalid), Result(x=unstd(arxvalid[order[0]]), fun=arfit[order[0]], nfev=nfev, nit=it))
        if status is not None:
            break
    return _final(unstd(arxvalid[order[0]]), arfit[order[0]], status, nfev, it, hist)


def run_de_sharded(fobj, lower, upper, stream, world, maxiter=100, popsize=10, mutation=0.5, recombination=0.9,
                   strategy="best1bin", xtol=1e-8, ftol=1e-8, constraints=None, **_ignored):
    """The multi-GPU semantics of the build (NOT a reference algorithm):
    `world` row shards -- blocks of ceil(P / world) rows, the last one short (stochopy_amd/parallel.py shard_bounds) --,
    donors drawn inside the shard, counters keyed by the global row, one global best per generation.  Simulated in one
    process."""
    n = len(lower)
    P = popsize
    Pc = -(-P // world)
    k = DONORS[strategy]
    X = latin_hypercube(stream, P, n, lower, upper)
    fit = fobj(X)
    g = int(np.argmin(fit))
    gfit, gbest = fit[g], X[g].copy()
    trace = [gfit]
    it = 1
    while True:
        it += 1
        U = np.empty_like(X)
        for r in range(world):
            sl = slice(r * Pc, min((r + 1) * Pc, P))
            draws = stream.de_generation(it, sl.stop - sl.start, n, k, (lower, upper) if constraints == "Random" else None,
                                         row0=r * Pc)
            U[sl] = de_candidates(X[sl], gbest, draws, mutation, recombination, strategy, lower, upper, constraints)
        gbest, gfit, status = greedy_select(it, U, fobj(U), gbest, X, fit, maxiter, xtol, ftol)
        trace.append(gfit)
        if status is not None:
            break
    res = _final(gbest, gfit, status, it * P, it, History(False, 0, 0, 0, 0))
    res["_trace"] = trace
    return res


RUNNERS = {"de": run_de, "pso": run_pso, "cpso": run_pso, "cmaes": run_cmaes, "vdcma": run_vdcma, "na": run_na}


def minimize(objective, bounds, x0=None, method="de", options=None, callback=None, rng="numpy-legacy"):
    """Oracle counterpart of stochopy.optimize.minimize (_helpers.py:44-94) for named objectives."""
    opts = dict(options or {})
    # (the reference's default: de/_de.py:27, cpso/_cpso.py:29 `updating="immediate"` -- rounds 1-5 defaulted to "deferred" here,
    #  a trap for a caller comparing the two packages' defaults: VERDICT r5)
    updating = opts.pop("updating", "immediate")
    if method in ("de", "pso", "cpso"):
        opts["updating"] = updating
    opts.pop("workers", None)
    opts.pop("backend", None)
    seed = opts.pop("seed", None)
    fobj = OBJECTIVES[objective] if isinstance(objective, str) else objective
    lower, upper = np.transpose(np.asarray(bounds, dtype=np.float64))
    stream = rng if hasattr(rng, "lhs_draws") else make_stream(rng, seed)
    if method == "pso":
        opts["competitivity"] = None
    elif method == "cpso":
        opts.setdefault("competitivity", 1.0)
    return RUNNERS[method](fobj, lower, upper, x0, stream, callback=callback, **opts)
