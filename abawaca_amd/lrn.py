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


def lrn_emit_rows(ctx: Context, values: np.ndarray, dps: np.ndarray) -> np.ndarray:
    """One cluster's rows the way ClusterEsomWriter prints them (reference src/ClusterWriter.cpp:92-98).
    values f64 [n_cols, ld] (dimension-major, column dp-1 = datapoint dp), dps int64 (1-based ids in row order)
    -> the rows as bytes (uint8 array)."""
    v = np.ascontiguousarray(values, dtype=np.float64)
    d = np.ascontiguousarray(dps, dtype=np.int64)
    n_cols, ld = v.shape
    n = C.c_uint64(0)
    rc = lib().abw_lrn_emit_rows(ctx.handle, C.c_void_p(v.ctypes.data), ld, n_cols, C.c_void_p(d.ctypes.data), len(d), None, 0, C.byref(n))
    if len(d) == 0:
        ctx.check(rc)
        return np.empty(0, dtype=np.uint8)
    if rc != -3:                                   # ABW_ERR_CAPACITY carries the size
        ctx.check(rc)
    text = np.empty(int(n.value), dtype=np.uint8)
    ctx.check(lib().abw_lrn_emit_rows(ctx.handle, C.c_void_p(v.ctypes.data), ld, n_cols, C.c_void_p(d.ctypes.data), len(d),
                                      C.c_void_p(text.ctypes.data), len(text), C.byref(n)))
    return text[:n.value]


def lrn_emit_rows_dev(ctx: Context, t_values, t_dps, t_text=None, t_row_off=None, resident=None):
    """Device-resident form: torch tensors values f64 [n_cols, ld], dps int64 [n_rows], text uint8 (or None: size only),
    row_off int64/uint64 [n_rows + 1] (or None).  t_values None + resident = (n_cols, ld): print from the copy
    lrn_table_load_dev keeps.  -> number of bytes the rows take."""
    n_cols, ld = t_values.shape if t_values is not None else resident
    n = C.c_uint64(0)
    rc = lib().abw_lrn_emit_rows_dev(ctx.handle, C.c_void_p(t_values.data_ptr()) if t_values is not None else None, ld, n_cols,
                                     C.c_void_p(t_dps.data_ptr()), t_dps.numel(),
                                     C.c_void_p(t_text.data_ptr()) if t_text is not None else None, t_text.numel() if t_text is not None else 0, C.byref(n),
                                     C.c_void_p(t_row_off.data_ptr()) if t_row_off is not None else None)
    if not (t_text is None and rc == -3):
        ctx.check(rc)
    return int(n.value)


def lrn_table_load_dev(ctx: Context, t_values):
    """Keeps a (transposed) copy of the device table f64 [n_cols, ld] for lrn_emit_rows_dev(ctx, None, ..., resident=shape);
    None drops it.  -> the shape to pass as `resident`."""
    if t_values is None:
        ctx.check(lib().abw_lrn_table_load_dev(ctx.handle, None, 0, 0))
        return None
    n_cols, ld = t_values.shape
    ctx.check(lib().abw_lrn_table_load_dev(ctx.handle, C.c_void_p(t_values.data_ptr()), ld, n_cols))
    return (n_cols, ld)


def lrn_emit_rows_dev_host_check(ctx: Context, dps: np.ndarray, resident) -> np.ndarray:
    """abw_lrn_emit_rows (host pointers) from the resident copy, whichever call loaded it: values = NULL."""
    n_cols, ld = resident
    d = np.ascontiguousarray(dps, dtype=np.int64)
    text = np.empty(len(d) * (23 * n_cols + 24), dtype=np.uint8)
    n = C.c_uint64(0)
    ctx.check(lib().abw_lrn_emit_rows(ctx.handle, None, ld, n_cols, C.c_void_p(d.ctypes.data), len(d), C.c_void_p(text.ctypes.data), len(text), C.byref(n)))
    return text[:n.value]
