"""Host-side entry to the GPU .lrn loader (abw_lrn_header / abw_lrn_parse): the reference's
ClusterData::ClusterData(esom_lrn_file, scaf_db), src/ClusterData.cpp:27-168, for tables written by abawaca-build."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import AbwError, Context, lib


def lrn_header(text: np.ndarray):
    """-> (ndps, ndims, [dimension names])"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    ndps, ndims = C.c_int64(0), C.c_int32(0)
    buf = C.create_string_buffer(max(16, len(text[:1 << 20]) + 16))
    rc = lib().abw_lrn_header(C.c_void_p(text.ctypes.data), len(text), C.byref(ndps), C.byref(ndims), buf, len(buf))
    if rc != 0:
        raise AbwError(rc, "abw_lrn_header: malformed header (reference src/ClusterData.cpp:36-111)")
    names = buf.raw.split(b"\0")[:ndims.value]
    return int(ndps.value), int(ndims.value), [n.decode(errors="replace") for n in names]


def lrn_parse(ctx: Context, text: np.ndarray):
    """-> (dp_names int32[ndps], keys int32[ndims, ndps] thousandths, dimension names)"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    ndps, ndims, names = lrn_header(text)
    dp_names = np.zeros(max(ndps, 1), dtype=np.int32)
    keys = np.zeros((ndims, max(ndps, 1)), dtype=np.int32)
    ctx.check(lib().abw_lrn_parse(ctx.handle, C.c_void_p(text.ctypes.data), len(text), ndps, ndims,
                                  C.c_void_p(dp_names.ctypes.data), C.c_void_p(keys.ctypes.data)))
    return dp_names[:ndps], keys[:, :ndps], names


def lrn_emit(ctx: Context, milli_rows: np.ndarray, first_key: int = 1) -> np.ndarray:
    """milli_rows int32 [n_rows, n_cols] -> the data rows of the .lrn as bytes (uint8 array)"""
    m = np.ascontiguousarray(milli_rows, dtype=np.int32)
    n_rows, n_cols = m.shape
    cap = n_rows * (n_cols * 8 + 12) + 64
    n = C.c_uint64(0)
    while True:
        text = np.empty(cap, dtype=np.uint8)
        rc = lib().abw_lrn_emit(ctx.handle, C.c_void_p(m.ctypes.data), n_rows, n_cols, int(first_key), C.c_void_p(text.ctypes.data), cap, C.byref(n))
        if rc == -3 and n.value > cap:            # ABW_ERR_CAPACITY: now we know the size
            cap = int(n.value)
            continue
        ctx.check(rc)
        return text[:n.value]
