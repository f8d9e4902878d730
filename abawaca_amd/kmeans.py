"""Host-side entry to the k-means restart extension (no counterpart in the reference; semantics in
DESIGN.md "K-means").  numpy in/out via abw_km_run; torch device tensors via abw_km_run_dev."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, lib


def tinymt_selftest(ctx: Context, seed: int, n: int):
    out = np.empty(n, dtype=np.uint32)
    ctx.check(lib().abw_km_tinymt_selftest(ctx.handle, int(seed) & 0xFFFFFFFF, n, C.c_void_p(out.ctypes.data)))
    return [int(v) for v in out]


def kmeans_restarts(ctx: Context, x_dn: np.ndarray, R: int, base_seed: int, iters: int, chunk: int = 0):
    """x_dn: fp32 [d, N] dimension-major.  -> (assign u8[R,N], centroids f32[R,2,d], inertia f64[R])"""
    x = np.ascontiguousarray(x_dn, dtype=np.float32)
    d, N = x.shape
    assign = np.empty((R, N), dtype=np.uint8)
    cent = np.empty((R, 2, d), dtype=np.float32)
    inertia = np.empty(R, dtype=np.float64)
    ctx.check(lib().abw_km_run(ctx.handle, C.c_void_p(x.ctypes.data), N, d, R, int(base_seed) & 0xFFFFFFFF, iters, chunk,
                               C.c_void_p(assign.ctypes.data), C.c_void_p(cent.ctypes.data), C.c_void_p(inertia.ctypes.data)))
    return assign, cent, inertia


def auto_chunk(ctx: Context, N: int, R: int) -> int:
    """The chunk length the library picks for chunk = 0 on this device (abw_km_auto_chunk)."""
    return int(lib().abw_km_auto_chunk(ctx.handle, int(N), int(R)))


def kmeans_restarts_dev(ctx: Context, t_x, N: int, d: int, R: int, base_seed: int, iters: int, chunk: int = 0,
                        t_assign=None, t_centroids=None, t_inertia=None):
    """Device-resident variant: t_x is a torch float32 tensor of shape [d, N] on the context's GPU.
    Asynchronous on the context's stream."""
    def p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None
    ctx.check(lib().abw_km_run_dev(ctx.handle, p(t_x), int(N), int(d), int(R), int(base_seed) & 0xFFFFFFFF, int(iters),
                                   int(chunk), p(t_assign), p(t_centroids), p(t_inertia)))
