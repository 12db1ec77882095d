"""GPU parity: the hand-written symmetric eigensolver (csrc/sx_eigh.hip, parallel block Jacobi on the fp64 matrix
cores) against numpy.linalg.eigh -- the reference's call at cmaes/_cmaes.py:303-305 -- and against the oracle's
canonical-sign form of it."""
import numpy as np
import pytest

from oracle import engine as oe

pytestmark = pytest.mark.gpu

# tolerances (relative to |C|): eigenvalues, reconstruction, orthogonality
EIG_RTOL = 1e-12
RESID_TOL = 1e-13
ORTH_TOL = 1e-13


@pytest.fixture(scope="module")
def ctx():
    from stochopy_amd import _device

    return _device.Context()


def cma_like(n, gens, rs):
    """covariances as the CMA-ES update produces them from C = I (cmaes/_cmaes.py:290-295)."""
    k = oe.cma_constants(n, 2 * n, 0.5)
    C = np.eye(n)
    for _ in range(gens):
        d, B = np.linalg.eigh(C)
        Y = (rs.randn(k["mu"], n) * np.sqrt(d)) @ B.T
        C = (1.0 - k["c1"] - k["cmu"]) * C + k["cmu"] * (Y.T * k["w"]) @ Y
        C = np.triu(C) + np.triu(C, 1).T
    return C


def make(kind, n, rs):
    if kind == "spd":
        A = rs.randn(n, n)
        return A @ A.T / n + 0.1 * np.eye(n)
    if kind == "cma":
        return cma_like(n, 3, rs)
    if kind == "indefinite":
        A = rs.randn(n, n)
        return A + A.T
    if kind == "graded":
        Q = np.linalg.qr(rs.randn(n, n))[0]
        return (Q * np.logspace(0, -8, n)) @ Q.T
    if kind == "repeated":
        Q = np.linalg.qr(rs.randn(n, n))[0]
        lam = np.repeat(rs.uniform(1, 2, (n + 3) // 4), 4)[:n]
        return (Q * lam) @ Q.T
    if kind == "diagonal":
        return np.diag(rs.uniform(-1, 1, n))
    if kind == "tiny":
        A = rs.randn(n, n)
        return (A @ A.T / n) * 1e-120
    if kind == "huge":
        A = rs.randn(n, n)
        return (A + A.T) * 1e120
    raise KeyError(kind)


def run(ctx, Cm, **kw):
    from stochopy_amd.linalg import Eigh

    n = Cm.shape[0]
    eig = Eigh(ctx, n)
    w, B = eig(ctx.upload(Cm), **kw)
    sweeps, conv, off = eig.info()
    return w.cpu().numpy(), B.cpu().numpy(), sweeps, conv, off


def check(Cm, w, B, sweeps, conv, EIG_RTOL=EIG_RTOL, RESID_TOL=RESID_TOL, ORTH_TOL=ORTH_TOL):
    n = Cm.shape[0]
    Cs = np.triu(Cm) + np.triu(Cm, 1).T
    wr, Br = oe.eigh_canonical(Cs)
    scale = max(np.abs(wr).max(), np.finfo(float).tiny)
    assert conv, f"not converged after {sweeps} sweeps"
    assert sweeps <= 30
    assert np.all(np.diff(w) >= 0.0)
    assert np.abs(w - wr).max() <= EIG_RTOL * scale
    nC = max(np.linalg.norm(Cs), np.finfo(float).tiny)
    assert np.linalg.norm(Cs / nC - (B * (w / nC)) @ B.T) <= RESID_TOL * max(1.0, np.sqrt(n) / 4)
    assert np.abs(B.T @ B - np.eye(n)).max() <= ORTH_TOL
    # the sign rule: the largest-magnitude component of every column is positive
    top = np.argmax(np.abs(B), axis=0)
    assert np.all(B[top, np.arange(n)] > 0.0)
    return wr, Br


SHAPES = [1, 2, 3, 7, 16, 17, 31, 32, 33, 50, 64, 65, 100, 128, 130, 200, 256, 300, 512]


@pytest.mark.parametrize("n", SHAPES)
@pytest.mark.parametrize("kind", ["spd", "cma", "indefinite"])
def test_eigh_vs_lapack(ctx, kind, n):
    rs = np.random.RandomState(1000 + n)
    Cm = make(kind, n, rs)
    w, B, sweeps, conv, off = run(ctx, Cm)
    wr, Br = check(Cm, w, B, sweeps, conv)
    # eigenvectors of well-separated eigenvalues agree with LAPACK's (canonical sign) to rounding / gap
    if n > 1:
        gap = np.minimum(np.diff(wr, prepend=-np.inf), np.diff(wr, append=np.inf)) / max(np.abs(wr).max(), 1e-300)
        good = gap > 1e-6
        if good.any():
            assert np.abs(B[:, good] - Br[:, good]).max() <= 1e-8


@pytest.mark.parametrize("n", [5, 48, 64, 96, 192, 384])
@pytest.mark.parametrize("kind", ["graded", "repeated", "diagonal", "tiny", "huge"])
def test_eigh_hard_spectra(ctx, kind, n):
    rs = np.random.RandomState(77 + n)
    Cm = make(kind, n, rs)
    w, B, sweeps, conv, off = run(ctx, Cm, max_sweeps=40)  # graded spectra in a random basis: ~20 sweeps
    check(Cm, w, B, sweeps, conv)


@pytest.mark.parametrize("n", [65, 128, 200, 512])
@pytest.mark.parametrize("kind", ["cma", "graded", "repeated"])
def test_eigh_warm_start(ctx, kind, n):
    """A starting basis (the previous generation's eigenvectors in the CMA-ES loop) changes the number of sweeps, not
    the result: the decomposition of a slightly different matrix started from B equals the cold one to rounding;
    a basis that is only nearly orthonormal is repaired (Newton-Schulz) rather than trusted."""
    import torch

    from stochopy_amd.linalg import Eigh

    rs = np.random.RandomState(400 + n)
    C0 = make(kind, n, rs)
    E = rs.randn(n, n) * 2e-3 * np.abs(C0).max() / np.sqrt(n)
    C1 = C0 + 0.5 * (E + E.T) if kind != "repeated" else C0 * 1.01
    eig = Eigh(ctx, n)
    with torch.cuda.stream(ctx.stream):  # torch ops and the solver's kernels on ONE stream
        _, B0 = eig(ctx.upload(C0), max_sweeps=40)
        start = B0.clone()
        start += 1e-9 * ctx.upload(rs.randn(n, n))  # not quite orthonormal any more
        d1 = ctx.upload(C1)
        wc, Bc = eig(d1, max_sweeps=40)
        wc, Bc = wc.cpu().numpy(), Bc.cpu().numpy()
        cold = eig.info()
        ww, Bw = eig(d1, max_sweeps=40, start=start)
        ww, Bw = ww.cpu().numpy(), Bw.cpu().numpy()
        warm = eig.info()
        # the output may alias the starting basis
        Bio = start.clone()
        eig(d1, B=Bio, max_sweeps=40, start=Bio)
        Bio = Bio.cpu().numpy()
    check(C1, ww, Bw, warm[0], warm[1])
    assert np.abs(ww - wc).max() <= EIG_RTOL * np.abs(wc).max()
    assert warm[0] <= cold[0]
    if kind == "graded":  # a graded spectrum in a random basis costs ~20 sweeps from scratch; the perturbation here is
        assert warm[0] <= cold[0] - 4  # unstructured (2e-3 |C|, far above the small eigenvalues), so not just 2 or 3
    assert np.array_equal(Bio, Bw)
    print(kind, n, "sweeps cold", cold[0], "warm", warm[0])


def test_eigh_reads_the_upper_triangle_only(ctx):
    """cmaes/_cmaes.py:303 mirrors the upper triangle before the decomposition; so does the kernel."""
    rs = np.random.RandomState(5)
    for n in (9, 150):
        Cm = make("spd", n, rs)
        dirty = Cm + np.tril(rs.randn(n, n), -1)
        w0, B0, *_ = run(ctx, Cm)
        w1, B1, *_ = run(ctx, dirty)
        assert np.array_equal(w0, w1) and np.array_equal(B0, B1)


def test_eigh_n1024(ctx):
    rs = np.random.RandomState(3)
    Cm = make("cma", 1024, rs)
    w, B, sweeps, conv, off = run(ctx, Cm, max_sweeps=24)
    check(Cm, w, B, sweeps, conv)


def test_eigh_c4_golden_matrices(ctx):
    """The covariance matrices of the BASELINE config-4 run (CMA-ES Rosenbrock n=512 P=1024 seed 0), generated by
    the oracle: eigenvalues rtol 1e-12, reconstruction and orthogonality 1e-13."""
    import oracle

    mats = []
    orig = oe.cma_covariance

    def spy(*a, **k):
        out = orig(*a, **k)
        mats.append(out.copy())
        return out

    oe.cma_covariance = spy
    try:
        oracle.minimize("rosenbrock", [[-5.12, 5.12]] * 512, method="cmaes",
                        options={"maxiter": 3, "popsize": 1024, "seed": 0, "sigma": 0.1})
    finally:
        oe.cma_covariance = orig
    assert len(mats) == 3
    for Cm in mats:
        w, B, sweeps, conv, off = run(ctx, Cm)
        wr, Br = check(Cm, w, B, sweeps, conv)
        assert np.allclose(w, wr, rtol=1e-12, atol=0)


@pytest.mark.parametrize("kind,n", [("cma", 100), ("cma", 130), ("spd", 200), ("indefinite", 256), ("cma", 257), ("spd", 384),
                                    ("cma", 512), ("graded", 192), ("repeated", 192), ("diagonal", 96)])
def test_eigh_with_the_refinement_step(ctx, kind, n):
    """sx_eigh_set_refine(1): once what the sweeps left is first order against every gap (off <= 1e-7 |C|, max |K| <= 1e-3,
    max |K| * off <= 1e-12 |C|; measured on the device) the last sweep is replaced by V <- V (I + K + K^2/2) and the
    second-order term of the eigenvalues -- what the CMA-ES loops run with.  Never more sweeps than without it; the
    residual stays at the 1e-12 |C| level (the rule's bound), orthogonality at rounding level; eigenvectors of separated
    eigenvalues agree with LAPACK's; a multiple eigenvalue (`repeated`) does not trip the step up."""
    from stochopy_amd import _lib

    L = _lib.lib()
    rs = np.random.RandomState(77 + n)
    Cm = make(kind, n, rs)
    w0, B0, sweeps0, conv0, _ = run(ctx, Cm)
    prev = L.sx_eigh_set_refine(1)
    try:
        from stochopy_amd.linalg import Eigh

        eig = Eigh(ctx, n)
        w, B = eig(ctx.upload(Cm))
        sweeps, conv, off = eig.info()
        refined = int(eig.ws[124:125].cpu().numpy().view(np.int32)[0])  # EighInfo.refine
        w, B = w.cpu().numpy(), B.cpu().numpy()
    finally:
        L.sx_eigh_set_refine(prev)
    assert sweeps <= sweeps0 and (not refined or sweeps < sweeps0 or sweeps0 == 0)
    wr, Br = check(Cm, w, B, sweeps, conv, EIG_RTOL=2e-12, RESID_TOL=2e-12, ORTH_TOL=1e-13)
    if n > 64 and kind in ("cma", "spd", "indefinite"):
        assert refined == 1  # generic spectra always reach the rule one sweep early
    gap = np.minimum(np.diff(wr, prepend=-np.inf), np.diff(wr, append=np.inf)) / max(np.abs(wr).max(), 1e-300)
    good = gap > 1e-6
    if good.any():
        assert np.abs(B[:, good] - Br[:, good]).max() <= 1e-8


@pytest.mark.parametrize("kind,n,warm", [("cma", 33, False), ("spd", 64, False), ("cma", 65, True), ("indefinite", 128, False),
                                         ("cma", 200, True), ("graded", 192, False), ("repeated", 192, False),
                                         ("cma", 512, True), ("spd", 512, False), ("cma", 1024, True)])
def test_eigh_resident_launch_is_the_launch_per_round_run(ctx, kind, n, warm):
    """Round 6: all rounds of a run inside ONE resident launch (sx_eigh_set_flow(1), the default: pair workgroups hand their
    rotations on through agent-scope words, tile workgroups follow behind counters) against one launch per round
    (sx_eigh_set_flow(0)): the same arithmetic on the same operands -- eigenvalues, eigenvectors and the run record equal
    bit for bit, with and without a warm start and the refinement step."""
    from stochopy_amd import _lib

    L = _lib.lib()
    rs = np.random.RandomState(1000 + n)
    Cm = make(kind, n, rs)
    kw = {}
    if warm:
        Cs = np.triu(Cm) + np.triu(Cm, 1).T
        E = rs.randn(n, n) * 1e-3
        _, V = np.linalg.eigh(Cs + (E + E.T) * np.abs(Cs).max())
        kw["start"] = ctx.upload(np.ascontiguousarray(V))
    outs = []
    prev = L.sx_eigh_set_flow(-2)
    try:
        for refine in (False, True):
            for mode in (0, 1, 1):
                L.sx_eigh_set_flow(mode)
                outs.append((refine, mode) + run(ctx, Cm, refine=refine, **kw))
    finally:
        L.sx_eigh_set_flow(prev)
    for k in range(0, len(outs), 3):
        ref = outs[k]
        check(Cm, *ref[2:6])
        for got in outs[k + 1:k + 3]:
            assert got[4:6] == ref[4:6], (got[:2], got[4:6], ref[4:6])  # sweeps, converged
            assert np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3]), (kind, n, got[:2])
            assert abs(got[6] - ref[6]) <= 1e-12 * max(ref[6], 1e-300)  # (sums of atomics: the order of arrival)
