"""Device plumbing (PyTorch-ROCm): HBM buffers, the engine stream, plan upload.

PyTorch is used for memory, streams and torch.distributed only; all compute is
in the HIP library.
"""
import ctypes as C
import threading

import numpy as np

from . import _lib

_torch = None
_TLS = threading.local()
_ARENA_BYTES = 1 << 20  # pinned staging for upload_async: one arena per host thread and device (_TLS.arenas)


def torch():
    global _torch
    if _torch is None:
        import torch as _t

        _torch = _t
    return _torch


class NoDeviceError(RuntimeError):
    pass


def require_device(index=None):
    """Fail loudly when no MI355X is visible -- there is no CPU path."""
    t = torch()
    if not t.cuda.is_available():
        raise NoDeviceError("stochopy_amd: no ROCm device visible (torch.cuda.is_available() is False); "
                            "backend='hip' has no CPU fallback")
    _lib.lib()
    dev = t.device("cuda", t.cuda.current_device() if index is None else index)
    return dev


def ptr(tensor):
    return C.c_void_p(tensor.data_ptr()) if tensor is not None else C.c_void_p(None)


class Context:
    """One device + one non-default stream the engine launches on."""

    def __init__(self, device=None):
        t = torch()
        self.device = require_device(device)
        self.L = _lib.lib()
        # ONE engine stream per device and host thread, kept for the life of the process: torch's caching allocator
        # keeps its free blocks per stream, so a fresh stream per run meant fresh hipMallocs for every population buffer
        # (~0.35 ms of the 0.8 ms a whole minimize() call at the metric shape spent before its first generation)
        key = (self.device.index if self.device.index is not None else t.cuda.current_device())
        streams = getattr(_TLS, "streams", None)
        if streams is None:
            streams = _TLS.streams = {}
        if key not in streams:
            with t.cuda.device(self.device):
                streams[key] = t.cuda.Stream(device=self.device)
        self.stream = streams[key]

    @property
    def stream_ptr(self):
        return C.c_void_p(self.stream.cuda_stream)

    def empty(self, shape, dtype=None):
        t = torch()
        return t.empty(shape, dtype=dtype or t.float64, device=self.device)

    def zeros(self, shape, dtype=None):
        t = torch()
        return t.zeros(shape, dtype=dtype or t.float64, device=self.device)

    def upload(self, array, dtype=None):
        t = torch()
        a = np.array(array, order="C", copy=True)  # also normalises negative / zero strides
        x = t.from_numpy(a).to(self.device, non_blocking=False)
        return x if dtype is None else x.to(dtype)

    def upload_async(self, array):
        """Small host array -> device WITHOUT the blocking pageable copy of ``upload`` (~0.2 ms each through
        ``Tensor.to``): the bytes go through a pinned arena (per host thread and device) and an asynchronous copy on the
        CURRENT torch stream -- callers run inside ``torch.cuda.stream(ctx.stream)``, so kernels enqueued afterwards see it."""
        t = torch()
        a = np.ascontiguousarray(array)
        nbytes = a.nbytes
        if nbytes == 0 or nbytes > _ARENA_BYTES // 4:
            return self.upload(a)
        # The arena belongs to THIS host thread and THIS device: a slice is claimed, filled and its copy
        # enqueued by one thread, so no other thread's copy can still be waiting to be enqueued when the arena wraps, and
        # the device-wide synchronize at the wrap covers every stream a copy of this arena may have gone to.
        arenas = getattr(_TLS, "arenas", None)
        if arenas is None:
            arenas = _TLS.arenas = {}
        key = self.device.index if self.device.index is not None else t.cuda.current_device()
        if key not in arenas:
            arenas[key] = [t.empty(_ARENA_BYTES, dtype=t.uint8).pin_memory(), 0]
        arena = arenas[key]
        off = (arena[1] + 63) & ~63
        if off + nbytes > _ARENA_BYTES:  # wrap: every copy that read the arena so far must have landed
            if t.cuda.is_current_stream_capturing():
                raise RuntimeError("upload_async: the staging arena wrapped during a stream capture (a synchronisation "
                                   "is not allowed there); upload before capturing")
            t.cuda.synchronize(self.device)
            off = 0
        arena[1] = off + nbytes
        stage = arena[0][off:off + nbytes]
        stage.numpy()[:] = a.reshape(-1).view(np.uint8)
        dev = t.empty(a.shape, dtype=t.from_numpy(a.reshape(-1)[:0]).dtype, device=self.device)  # (ascontiguousarray: ndim >= 1)
        dev.reshape(-1).view(t.uint8).copy_(stage, non_blocking=True)
        return dev

    def sync(self):
        self.stream.synchronize()

    def read_state(self, state_tensor):
        """D2H of the 64-byte sx_state."""
        t = torch()
        with t.cuda.stream(self.stream):
            host = state_tensor.cpu()
        raw = host.numpy().tobytes()
        return _lib.SxState.from_buffer_copy(raw)


def evaluate(ctx, fun_id, X, n, f=None, xm=None, xstd=None, part=None):
    """f[i] = objective(X[i]) on the device (sx_eval)."""
    P = X.shape[0]
    if f is None:
        f = ctx.empty((P,))
    pf, pi = (part if part is not None else (None, None))
    ldx = X.stride(0) if P > 1 else max(X.stride(0), n)  # a length-1 axis may carry any stride
    _lib.check(ctx.L.sx_eval(fun_id, ptr(X), P, n, ldx, ptr(xm), ptr(xstd), ptr(f), ptr(pf),
                             ptr(pi), ctx.stream_ptr), "sx_eval")
    return f
