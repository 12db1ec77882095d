"""The eigensolver with and without the first-order refinement step (sx_eigh_set_refine): sweeps, residual, orthogonality,
eigenvalue and eigenvector agreement with LAPACK (canonical signs) on test matrices and on the covariance matrices of a
C4 run (warm-started from the previous generation's LAPACK eigenvectors, as the CMA-ES loop does).  usage: [n P gens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle
from oracle import engine as oe
from stochopy_amd import _device, _lib
from stochopy_amd.linalg import Eigh
from test_gpu_eigh import make

ctx = _device.Context()
L = _lib.lib()


def stats(Cm, w, B):
    n = len(Cm)
    Cs = np.triu(Cm) + np.triu(Cm, 1).T
    wr, Br = oe.eigh_canonical(Cs)
    nC = np.linalg.norm(Cs)
    return (np.abs(w - wr).max() / np.abs(wr).max(), np.linalg.norm(Cs / nC - (B * (w / nC)) @ B.T),
            np.abs(B.T @ B - np.eye(n)).max(), np.abs(B - Br).max())


def one(tag, Cm, start=None, tol=0.0):
    n = len(Cm)
    out = []
    for mode in (0, 1):
        L.sx_eigh_set_refine(mode)
        eig = Eigh(ctx, n)
        kw = {}
        if start is not None:
            kw["start"] = ctx.upload(start)
        w, B = eig(ctx.upload(Cm), tol=tol, **kw)
        sw, conv, off = eig.info()
        hdr = eig.ws[:2].cpu().numpy().view(np.int32)
        raw = eig.ws[:256].cpu().numpy()
        refined = int(raw[124:125].view(np.int32)[0])  # EighInfo.refine (byte 992)
        nrm2, offm, kmax2 = raw[2], raw[64:124], raw[125:185]
        left = ["%.0e/%.0e" % (np.sqrt(offm[k] / nrm2), np.sqrt(kmax2[k])) for k in range(max(0, sw - 2), sw)]
        e = stats(Cm, w.cpu().numpy(), B.cpu().numpy())
        out.append("mode %d: sweeps %2d conv %d refined %d | eig %.1e resid %.1e orth %.1e vec %.1e | off/maxK after the last two sweeps: %s"
                   % ((mode, sw, conv, refined) + e + (" ".join(left),)))
    L.sx_eigh_set_refine(-1)
    print(tag, "\n   " + "\n   ".join(out), flush=True)


rs = np.random.RandomState(7)
for kind, n in (("cma", 130), ("spd", 200), ("indefinite", 256), ("cma", 512), ("graded", 192), ("repeated", 192), ("spd", 1024)):
    one("%s n=%d" % (kind, n), make(kind, n, rs))
n, P, gens = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (512, 1024, 30)
rec = []
def record(C):
    w, V = np.linalg.eigh(C); rec.append((C.copy(), V)); return w, V
oracle.minimize("rosenbrock", [[-5.12, 5.12]] * n, method="cmaes",
                options={"popsize": P, "seed": 0, "maxiter": gens, "ftol": -1.0, "xtol": 0.0, "eigh": record}, rng="philox")
tol = max(1e-14, n * 1.1102230246251565e-16)
for g in range(1, len(rec)):
    if g in (1, 2, 5, 9) or g >= 11:
        one("C4-like generation %d (warm start)" % (g + 1), rec[g][0], start=rec[g - 1][1], tol=tol)
