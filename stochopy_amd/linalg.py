"""Device-side symmetric eigendecomposition (csrc/sx_eigh.hip) behind a small host handle.

Reference: stochopy/optimize/cmaes/_cmaes.py:303-305 (``C = triu(C) + triu(C,1).T; D, B = np.linalg.eigh(C)``,
ascending order).  The decomposition is a hand-written parallel block Jacobi (pivot sweeps in LDS, similarity
updates on the fp64 matrix cores); eigenvector signs follow a stated rule -- the component of largest magnitude
(lowest row on ties) is positive -- because LAPACK's signs are an accident of its internals that no other
solver can reproduce.
"""
import ctypes as C

from . import _device, _lib

__all__ = ["Eigh"]


class Eigh:
    """``eig = Eigh(ctx, n); w, B = eig(Cdev)``: eigenvalues ascending, eigenvectors in the columns of B.

    Everything is enqueued on ``ctx.stream``; nothing waits for the device.  ``info()`` synchronises and returns
    (sweeps, converged, off) of the last call, ``off`` = off-diagonal mass the last sweep left behind / |C|_F.
    """

    def __init__(self, ctx, n):
        t = _device.torch()
        self.ctx, self.n = ctx, int(n)
        self.bytes = int(ctx.L.sx_eigh_workspace_bytes(self.n))
        if self.bytes <= 0:
            raise ValueError(f"sx_eigh_workspace_bytes({n}) = {self.bytes}")
        self.ws = t.empty((self.bytes + 7) // 8, dtype=t.float64, device=ctx.device)
        with t.cuda.stream(ctx.stream):  # (the stream sx_eigh writes the record on: a memset on torch's current stream --
            self.ws[:256].zero_()        #  the default one for callers outside the engine's -- could land on a running call)
        # the run record: info() before the first decomposition reads zeros, not stale memory
        self.w = ctx.empty((self.n,))
        self.B = ctx.empty((self.n, self.n))

    def __call__(self, Cmat, w=None, B=None, max_sweeps=0, tol=0.0, start=None, refine=None):
        """``start``: optional (n, n) nearly orthonormal basis to start from (the previous decomposition's B; may be
        the output buffer itself).  ``refine``: True / False allows / forbids the first-order refinement step in place of
        the last sweep for THIS call (``sx_eigh_refined``: nothing process-wide is touched, so other host threads' runs keep
        their mode); None: the library's current mode (``sx_eigh_set_refine`` / ``SX_EIGH_REFINE``)."""
        n = self.n
        if tuple(Cmat.shape) != (n, n) or not Cmat.is_contiguous():
            raise ValueError(f"expected a contiguous ({n},{n}) device matrix")
        w = self.w if w is None else w
        B = self.B if B is None else B
        p = _device.ptr
        if start is not None and (tuple(start.shape) != (n, n) or not start.is_contiguous()):
            raise ValueError(f"start: expected a contiguous ({n},{n}) device matrix")
        _lib.check(self.ctx.L.sx_eigh_refined(p(Cmat), n, p(start), p(w), p(B), p(self.ws), self.bytes, int(max_sweeps),
                                              float(tol), -1 if refine is None else int(bool(refine)), self.ctx.stream_ptr),
                   "sx_eigh")
        return w, B

    def info(self):
        sweeps, conv, off = C.c_int(0), C.c_int(0), C.c_double(0.0)
        _lib.check(self.ctx.L.sx_eigh_info(_device.ptr(self.ws), C.byref(sweeps), C.byref(conv), C.byref(off),
                                           self.ctx.stream_ptr), "sx_eigh_info")
        return sweeps.value, bool(conv.value), off.value
