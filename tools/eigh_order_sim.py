"""Round 6, CPU only: how many ROUNDS does the block Jacobi method of csrc/sx_eigh.hip need on the covariance matrices of
BASELINE config 4 (warm-started from the previous generation's eigenvectors) under other orderings and pivot solvers?
A numpy model of the method (blocks of 16, 16 disjoint block pairs per round):
  order  cyclic = the round-robin tournament the kernel uses; dyn = greedy maximum-weight matching on the current block norms
         (dynamic ordering, Becka / Oksa / Vajtersic); stale = the same from the norms of one round earlier with that round's
         pairs zeroed (what a pipelined kernel could know)
  pivot  exact = the 32x32 pivot diagonalised exactly; jac = ONE cyclic Jacobi sweep over the pivot (cross pairs only except
         in every 31st round), which is what the kernel does
  sort   1 = the pivot's eigenvalues sorted ascending across the pair (smaller half to the lower block)
Counted: rounds until off(M) <= 2e-7 |C|_F (where the kernel's refinement step takes over).
usage: eigh_order_sim.py [generations, comma separated] [exact|jac]   (jac: minutes per line)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.optimize import linear_sum_assignment
import oracle

_n, _P, _gens = 512, 1024, 14
_rec = []
def _record(C):
    w, V = np.linalg.eigh(C); _rec.append((C.copy(), V)); return w, V
oracle.minimize("rosenbrock", [[-5.12, 5.12]] * _n, method="cmaes",
                options={"popsize": _P, "seed": 0, "maxiter": _gens, "ftol": -1.0, "xtol": 0.0, "eigh": _record}, rng="philox")
Cs, Vs = np.array([r[0] for r in _rec]), np.array([r[1] for r in _rec])
n = 512; bs = 16; nb = n // bs

def rr_pairs(r, m):
    q = m - 1
    out = [(q, r)]
    for k in range(1, m // 2):
        out.append(((r + k) % q, (r - k) % q))
    return out

def blockw(M):
    B = M.reshape(nb, bs, nb, bs)
    return np.einsum('aibj,aibj->ab', B, B)

def off2(M):
    return (M**2).sum() - (np.diag(M)**2).sum()

def jac_rot(app, aqq, apq):
    if apq == 0: return 1.0, 0.0
    d = aqq - app; h = 2 * apq
    r = np.hypot(d, h)
    c2 = abs(d) / r
    c = np.sqrt((1 + c2) / 2)
    s = (np.sign(d) if d != 0 else 1.0) * h / r / (2 * c)
    return c, s

def pivot_sweep(Pm, full):
    # systolic-equivalent: cyclic sweep; positions 0..15 block a, 16..31 block b
    m = 2 * bs
    A = Pm.copy(); W = np.eye(m)
    if full:
        rounds = [[(x, y) for (x, y) in rr_pairs(r, m)] for r in range(m - 1)]
    else:
        rounds = [[(i, bs + (i + r) % bs) for i in range(bs)] for r in range(bs)]
    for prs in rounds:
        J = np.eye(m)
        for (p, q) in prs:
            if p > q: p, q = q, p
            c, s = jac_rot(A[p, p], A[q, q], A[p, q])
            J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
        A = J.T @ A @ J; W = W @ J
    return A, W

def solve_pivot(Pm, mode, full):
    if mode == 'exact':
        w, Q = np.linalg.eigh(Pm)
        ri, ci = linear_sum_assignment(-np.abs(Q))
        Qn = np.empty_like(Q); Qn[:, ri] = Q[:, ci] * np.sign(Q[ri, ci])
        return Qn
    A, W = pivot_sweep(Pm, full)
    return W

def apply_round(M, pairs, mode, full, sort):
    U = np.eye(n)
    for (a, b) in pairs:
        if a > b: a, b = b, a
        idx = np.r_[a*bs:(a+1)*bs, b*bs:(b+1)*bs]
        Pm = M[np.ix_(idx, idx)]
        Q = solve_pivot(Pm, mode, full)
        if sort:
            dg = np.einsum('ij,ik,kj->j', Q, Pm, Q)
            Q = Q[:, np.argsort(dg, kind='stable')]
        U[np.ix_(idx, idx)] = Q
    return U.T @ M @ U

def greedy(W):
    W = W.copy()
    pairs = []
    used = np.zeros(nb, bool)
    iu = np.triu_indices(nb, 1)
    order = np.argsort(-W[iu])
    for o in order:
        a, b = iu[0][o], iu[1][o]
        if not used[a] and not used[b]:
            used[a] = used[b] = True
            pairs.append((a, b))
            if len(pairs) == nb // 2: break
    return pairs

def run(M0, order, mode, sort, target, fullevery=31):
    M = M0.copy()
    nrm2 = (M**2).sum()
    hist = []
    rounds = 0
    Wprev = None; prevpairs = None
    while rounds < 330:
        o = np.sqrt(max(off2(M), 0) / nrm2)
        if o <= target: break
        if order == 'cyclic':
            pairs = rr_pairs(rounds % (nb - 1), nb)
        elif order == 'dyn':
            pairs = greedy(blockw(M))
        elif order == 'stale':
            if Wprev is None:
                pairs = greedy(blockw(M))
            else:
                W = Wprev.copy()
                for (a, b) in prevpairs: W[a, b] = W[b, a] = 0
                pairs = greedy(W)
        Wprev = blockw(M); prevpairs = pairs
        full = (rounds % fullevery == 0)
        M = apply_round(M, pairs, mode, full, sort)
        rounds += 1
        if rounds % 31 == 0: hist.append(np.sqrt(max(off2(M), 0) / nrm2))
    return rounds, hist

gens = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,12").split(",")]
modes = (sys.argv[2] if len(sys.argv) > 2 else "exact").split(",")
for g in gens:
    C = Cs[g]; C = np.triu(C) + np.triu(C, 1).T
    V = Vs[g - 1]
    M0 = V.T @ C @ V
    for target in (2e-7,):
        for mode in modes:
            for order in ('cyclic', 'dyn', 'stale'):
                for sort in (False, True):
                    r, h = run(M0, order, mode, sort, target)
                    print("gen %d target %.0e %-6s %-7s sort=%d rounds %3d  per-31: %s" % (g + 1, target, mode, order, sort, r, " ".join("%.1e" % x for x in h)), flush=True)
