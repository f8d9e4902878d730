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


ARITH_F32, ARITH_BF16X3 = 0, 1


def set_arith(ctx: Context, arith: int) -> None:
    """What the Lloyd passes compute in from now on (abw_km_set_arith): ARITH_F32, the exact default, or ARITH_BF16X3, the
    three-plane bf16 split held to the documented Lloyd tolerance instead of bit parity."""
    ctx.check(lib().abw_km_set_arith(ctx.handle, int(arith)))


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


class _DevView:
    """Zero-copy torch view of a device buffer owned by the library (via __cuda_array_interface__)."""
    def __init__(self, ptr, count, typestr):
        self.__cuda_array_interface__ = dict(shape=(int(count),), typestr=typestr, data=(int(ptr), False), version=2)


class KmExchange(C.Structure):
    _fields_ = [("d_init", C.c_void_p), ("init_count", C.c_uint64), ("d_moments", C.c_void_p), ("moments_count", C.c_uint64),
                ("d_sums", C.c_void_p), ("sums_count", C.c_uint64)]


class ShardRun:
    """One shard of a point-sharded k-means run (abw_km_shard_*): the caller combines `init` (int32),
    `moments` and `sums` (fp64) over the shards between the phases -- abawaca_amd.multi has the
    collectives; tests drive several shards from one process."""

    def __init__(self, ctx: Context, t_x, N_local: int, first_point: int, N_total: int, d: int, R: int, base_seed: int, chunk: int):
        import torch
        self.ctx, self.R, self.N_local, self.d = ctx, R, N_local, d
        self._x = t_x                                         # keep the points alive
        ex = KmExchange()
        ctx.check(lib().abw_km_shard_begin(ctx.handle, C.c_void_p(t_x.data_ptr()), int(N_local), int(first_point), int(N_total),
                                           int(d), int(R), int(base_seed) & 0xFFFFFFFF, int(chunk), C.byref(ex)))
        dev = t_x.device
        self.init = torch.as_tensor(_DevView(ex.d_init, ex.init_count, "<i4"), device=dev)
        self.moments = torch.as_tensor(_DevView(ex.d_moments, ex.moments_count, "<f8"), device=dev)
        self.sums = torch.as_tensor(_DevView(ex.d_sums, ex.sums_count, "<f8"), device=dev)

    def ready(self):
        self.ctx.check(lib().abw_km_shard_ready(self.ctx.handle))

    def step(self, t_assign=None):
        self.ctx.check(lib().abw_km_shard_step(self.ctx.handle, C.c_void_p(t_assign.data_ptr()) if t_assign is not None else None))

    def update(self, last: bool):
        self.ctx.check(lib().abw_km_shard_update(self.ctx.handle, 1 if last else 0))

    def finish(self, t_centroids=None, t_inertia=None):
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        self.ctx.check(lib().abw_km_shard_finish(self.ctx.handle, p(t_centroids), p(t_inertia)))


def kmeans_sharded_dev(ctx: Context, t_x_local, first_point: int, N_total: int, R: int, base_seed: int, iters: int,
                       chunk: int, dist, t_assign=None, t_centroids=None, t_inertia=None):
    """Point-sharded run over torch.distributed (RCCL on GPUs): this rank holds points
    [first_point, first_point + N_local) as t_x_local [d, N_local] on the context's GPU.  Every rank ends
    with the same centroids / inertia and its own slice of the assignments."""
    import torch
    from . import multi
    d, N_local = t_x_local.shape
    run = ShardRun(ctx, t_x_local, N_local, first_point, N_total, d, R, base_seed, chunk)
    # library stream != torch's: order the two by hand; shared (ctx.use_torch_stream): the device orders them
    shared = ctx.shares_torch_stream

    def lib_done():
        if not shared:
            ctx.synchronize()

    def torch_done():
        if not shared:
            torch.cuda.synchronize(t_x_local.device)
    lib_done()
    multi.combine_int_sum(run.init, dist)
    multi.combine_ordered_sum(run.moments, dist, ctx)
    torch_done()
    run.ready()
    for it in range(iters + 1):
        last = it == iters
        run.step(t_assign if last else None)
        lib_done()
        multi.combine_ordered_sum(run.sums, dist, ctx)
        torch_done()
        run.update(last)
    run.finish(t_centroids, t_inertia)
    return run


# ---- the k-means separator on a resident feature table (abw_km_load / abw_km_node) --------------------

class KmColumnStore:
    """fp32 features of every datapoint, resident on the GPU for the whole recursion."""

    def __init__(self, ctx: Context, x_dn: np.ndarray, dp2scaf):
        self.ctx = ctx
        self.x = np.ascontiguousarray(x_dn, dtype=np.float32)
        self.D, self.N = self.x.shape
        self.dp2scaf = np.ascontiguousarray(dp2scaf, dtype=np.int32)
        self.S = int(self.dp2scaf.max()) + 1
        self.scaf_ndps = np.bincount(self.dp2scaf, minlength=self.S).astype(np.int32)
        self.scaf_first = np.concatenate([[0], np.cumsum(self.scaf_ndps)[:-1]]).astype(np.int64)
        ctx.check(lib().abw_km_load(ctx.handle, C.c_void_p(self.x.ctypes.data), self.D, self.N,
                                    C.c_void_p(self.dp2scaf.ctypes.data), self.S))


def km_node(store: KmColumnStore, scaf_ids, R: int, base_seed: int, iters: int, min_ndps: int = 100, chunk: int = 0):
    """One node (scaf_ids ascending, None = all).  chunk = 0: the library's choice for this device, reported in
    KmSplit.chunk.  -> (KmSplit, scaf_side u8[m], dp_raw_side u8[n])"""
    from ._lib import KmSplit
    ids = None if scaf_ids is None else np.ascontiguousarray(scaf_ids, dtype=np.int32)
    m = store.S if ids is None else len(ids)
    n = store.N if ids is None else int(store.scaf_ndps[ids].sum())
    out = KmSplit()
    side = np.zeros(max(m, 1), dtype=np.uint8)
    raw = np.zeros(max(n, 1), dtype=np.uint8)
    store.ctx.check(lib().abw_km_node(store.ctx.handle, C.c_void_p(ids.ctypes.data) if ids is not None else None, m, int(R),
                                      int(base_seed) & 0xFFFFFFFF, int(iters), int(chunk), int(min_ndps), C.byref(out),
                                      C.c_void_p(side.ctypes.data), C.c_void_p(raw.ctypes.data)))
    return out, side[:m], raw[:n]
