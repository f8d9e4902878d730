"""Host-side mirror of the reference's signature stage (Scaf / Scaf_segment, reference
src/abawaca-build.cpp:34-62,103-228) over the C ABI.  numpy in, numpy out; device-pointer variants
take torch tensors (used by bench.py so that inputs are resident in HBM before timing)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import ABW_ERR_CAPACITY, AbwError, Context, SIG_NCOLS, SIG_NDIMS, lib


def dim_names():
    """The 180 dimension names in the reference's order (dimension_order, abawaca-build.cpp:69)."""
    buf = C.create_string_buffer(1024)
    rc = lib().abw_sig_dim_names(buf, 1024)
    if rc:
        raise AbwError(rc, "abw_sig_dim_names")
    return [s.decode() for s in buf.raw.split(b"\0")[:SIG_NDIMS]]


def _ptr(a):
    return C.c_void_p(a.ctypes.data) if a is not None else None


def segment_host(ascii_blob: np.ndarray, scaf_off: np.ndarray, window: int = 2000):
    """Scaf::Scaf's segmentation on the host -> (seg_scaf u32, seg_start i64, seg_end i64), 1-based inclusive."""
    blob = np.ascontiguousarray(ascii_blob, dtype=np.uint8)
    off = np.ascontiguousarray(scaf_off, dtype=np.uint64)
    n_scaf = len(off) - 1
    cap = int(len(blob) // window + n_scaf + 16)
    while True:
        seg_scaf = np.empty(cap, dtype=np.uint32)
        st = np.empty(cap, dtype=np.uint64)
        en = np.empty(cap, dtype=np.uint64)
        n = C.c_uint64(0)
        rc = lib().abw_sig_segment_host(_ptr(blob), _ptr(off), n_scaf, window, cap, _ptr(seg_scaf), _ptr(st), _ptr(en), C.byref(n))
        if rc == ABW_ERR_CAPACITY:
            cap = int(n.value)
            continue
        if rc:
            raise AbwError(rc, "abw_sig_segment_host (lower-case 'n' in the input?)" if rc == -4 else "abw_sig_segment_host")
        k = int(n.value)
        return seg_scaf[:k], st[:k].astype(np.int64), en[:k].astype(np.int64)


def build_signatures(ctx: Context, ascii_blob: np.ndarray, scaf_off: np.ndarray, window: int = 2000,
                     want_dims: bool = True, want_milli: bool = True):
    """Whole stage through abw_sig_build.
    -> dict(seg_scaf, seg_start, seg_end, dims f64[n,180] | None, milli i32[n,179] | None)"""
    blob = np.ascontiguousarray(ascii_blob, dtype=np.uint8)
    off = np.ascontiguousarray(scaf_off, dtype=np.uint64)
    n_scaf = len(off) - 1
    cap = int(len(blob) // window + n_scaf + 16)
    while True:
        seg_scaf = np.empty(cap, dtype=np.uint32)
        st = np.empty(cap, dtype=np.uint64)
        en = np.empty(cap, dtype=np.uint64)
        dims = np.empty((cap, SIG_NDIMS), dtype=np.float64) if want_dims else None
        milli = np.empty((cap, SIG_NCOLS), dtype=np.int32) if want_milli else None
        n = C.c_uint64(0)
        rc = lib().abw_sig_build(ctx.handle, _ptr(blob), _ptr(off), n_scaf, window, cap, _ptr(seg_scaf), _ptr(st), _ptr(en),
                                 _ptr(dims), _ptr(milli), C.byref(n))
        if rc == ABW_ERR_CAPACITY:
            cap = int(n.value)
            continue
        ctx.check(rc)
        k = int(n.value)
        return dict(seg_scaf=seg_scaf[:k], seg_start=st[:k].astype(np.int64), seg_end=en[:k].astype(np.int64),
                    dims=dims[:k] if want_dims else None, milli=milli[:k] if want_milli else None)


# ---- device-pointer variants (torch tensors on the context's GPU) ------------------------------
def segment_dev(ctx: Context, t_ascii, total_len: int, t_scaf_off, window, t_seg_scaf, t_seg_start, t_seg_end) -> int:
    n = C.c_uint64(0)
    cap = int(t_seg_scaf.numel())
    ctx.check(lib().abw_sig_segment_dev(ctx.handle, C.c_void_p(t_ascii.data_ptr()), int(total_len),
                                        C.c_void_p(t_scaf_off.data_ptr()), int(t_scaf_off.numel() - 1), int(window), cap,
                                        C.c_void_p(t_seg_scaf.data_ptr()), C.c_void_p(t_seg_start.data_ptr()),
                                        C.c_void_p(t_seg_end.data_ptr()), C.byref(n)))
    return int(n.value)


def segment_static(ctx: Context, t_ascii, total_len: int, t_scaf_off, window, t_seg_scaf, t_seg_start, t_seg_end, t_result):
    """abw_sig_segment_static: no allocation, no synchronisation; t_result (int64[2] on the device) receives the number
    of segments and the lower-case-n flag."""
    ctx.check(lib().abw_sig_segment_static(ctx.handle, C.c_void_p(t_ascii.data_ptr()), int(total_len),
                                           C.c_void_p(t_scaf_off.data_ptr()), int(t_scaf_off.numel() - 1), int(window),
                                           int(t_seg_scaf.numel()), C.c_void_p(t_seg_scaf.data_ptr()),
                                           C.c_void_p(t_seg_start.data_ptr()), C.c_void_p(t_seg_end.data_ptr()),
                                           C.c_void_p(t_result.data_ptr())))


def count_dev(ctx: Context, t_ascii, total_len: int, t_scaf_off, t_seg_scaf, t_seg_start, t_seg_end, n_seg: int,
              t_dims=None, t_milli=None, t_counts=None):
    def p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None
    ctx.check(lib().abw_sig_count_dev(ctx.handle, p(t_ascii), int(total_len), p(t_scaf_off), p(t_seg_scaf), p(t_seg_start),
                                      p(t_seg_end), int(n_seg), p(t_dims), p(t_milli), p(t_counts)))
