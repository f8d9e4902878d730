"""ctypes binding of oracle/liboracle.so (the CPU restatement).  Test infrastructure only:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by abawaca_amd."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class DimBest(C.Structure):
    _fields_ = [("found", C.c_int32), ("key", C.c_int32), ("sensitivity", C.c_double), ("specificity", C.c_double)]


class Split(C.Structure):
    _fields_ = [("split", C.c_int32), ("dimension", C.c_int32), ("key", C.c_int32), ("legal", C.c_int32),
                ("sensitivity", C.c_double), ("specificity", C.c_double), ("ndps1", C.c_int64), ("ndps2", C.c_int64)]


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True,
                           stdout=subprocess.DEVNULL)
        _LIB = C.CDLL(path)
        _LIB.abw_or_segments.restype = C.c_uint64
        _LIB.abw_or_milli.restype = C.c_int32
        _LIB.abw_or_milli.argtypes = [C.c_double]
        _LIB.abw_or_tinymt_next.restype = C.c_uint32
        _LIB.abw_or_sam_parse.restype = C.c_int64
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def mer_table():
    out = np.empty(340, dtype=np.int32)
    lib().abw_or_mer_table(_p(out, C.c_int32))
    return out


def upcase(seq: np.ndarray):
    s = np.ascontiguousarray(seq, dtype=np.uint8).copy()
    rc = lib().abw_or_upcase(_p(s, C.c_uint8), C.c_uint64(len(s)))
    return rc, s


def segments(seq: np.ndarray, window: int = 2000):
    s = np.ascontiguousarray(seq, dtype=np.uint8)
    cap = max(1, len(s))
    st = np.empty(cap, dtype=np.uint64)
    en = np.empty(cap, dtype=np.uint64)
    n = lib().abw_or_segments(_p(s, C.c_uint8), C.c_uint64(len(s)), C.c_uint32(window),
                              _p(st, C.c_uint64), _p(en, C.c_uint64), C.c_uint64(cap))
    return st[:n].astype(np.int64), en[:n].astype(np.int64)


def signature(seg: np.ndarray):
    s = np.ascontiguousarray(seg, dtype=np.uint8)
    dims = np.empty(180, dtype=np.float64)
    counts = np.empty(340, dtype=np.uint32)
    totals = np.empty(4, dtype=np.uint32)
    lib().abw_or_signature(_p(s, C.c_uint8), C.c_uint64(len(s)), _p(dims, C.c_double),
                           _p(counts, C.c_uint32), _p(totals, C.c_uint32))
    return dims, counts, totals


def milli_of(dims: np.ndarray):
    return np.array([lib().abw_or_milli(float(x)) for x in dims], dtype=np.int32)


def build_signatures(ascii_blob: np.ndarray, scaf_off: np.ndarray, window: int = 2000):
    """Whole signature stage on the CPU: -> seg_scaf, seg_start, seg_end (1-based incl.), dims [n,180]."""
    seg_scaf, seg_start, seg_end, rows = [], [], [], []
    for i in range(len(scaf_off) - 1):
        rc, s = upcase(ascii_blob[scaf_off[i]:scaf_off[i + 1]])
        if rc != 0:
            raise ValueError("lower-case n in scaffold %d" % i)
        st, en = segments(s, window)
        for a, b in zip(st, en):
            seg_scaf.append(i)
            seg_start.append(int(a))
            seg_end.append(int(b))
            rows.append(signature(s[a - 1:b])[0])
    dims = np.array(rows, dtype=np.float64).reshape(len(rows), 180)
    return (np.array(seg_scaf, dtype=np.int32), np.array(seg_start, dtype=np.int64),
            np.array(seg_end, dtype=np.int64), dims)


def scg_csr(scg_scaf, scg_id, S):
    order = np.argsort(scg_scaf, kind="stable")
    off = np.zeros(S + 1, dtype=np.int32)
    np.add.at(off, np.asarray(scg_scaf, dtype=np.int64) + 1, 1)
    off = np.cumsum(off).astype(np.int32)
    return off, np.ascontiguousarray(np.asarray(scg_id, dtype=np.int32)[order])


def separate(keys, dp2scaf, scaf_ndps, scaf_len, scg_off, scg_ids, n_scg, node_dps, min_ndps=100, nthreads=0):
    """keys int32 [D, N].  -> (Split, per_dim recarray, dp_raw_side uint8[n], scaf_side uint8[S])"""
    keys = np.ascontiguousarray(keys, dtype=np.int32)
    D, N = keys.shape
    S = len(scaf_ndps)
    dp2scaf = np.ascontiguousarray(dp2scaf, dtype=np.int32)
    scaf_ndps = np.ascontiguousarray(scaf_ndps, dtype=np.int32)
    scaf_len = np.ascontiguousarray(scaf_len, dtype=np.int64)
    scg_off = np.ascontiguousarray(scg_off, dtype=np.int32)
    scg_ids = np.ascontiguousarray(scg_ids, dtype=np.int32)
    node_dps = np.ascontiguousarray(node_dps, dtype=np.int32)
    n = len(node_dps)
    out = Split()
    per = (DimBest * D)()
    raw = np.zeros(n, dtype=np.uint8)
    side = np.zeros(S, dtype=np.uint8)
    rc = lib().abw_or_separate(_p(keys, C.c_int32), C.c_int64(N), C.c_int32(D), _p(dp2scaf, C.c_int32),
                               _p(scaf_ndps, C.c_int32), _p(scaf_len, C.c_int64), C.c_int32(S),
                               _p(scg_off, C.c_int32), _p(scg_ids, C.c_int32), C.c_int32(n_scg),
                               _p(node_dps, C.c_int32), C.c_int64(n), C.c_int32(min_ndps), C.c_int(nthreads),
                               C.byref(out), per, _p(raw, C.c_uint8), _p(side, C.c_uint8))
    if rc != 0:
        raise RuntimeError("abw_or_separate failed: %d" % rc)
    per_dim = np.array([(p.found, p.key, p.sensitivity, p.specificity) for p in per],
                       dtype=[("found", "i4"), ("key", "i4"), ("sens", "f8"), ("spec", "f8")])
    return out, per_dim, raw, side


def recursive_bins(keys, dp2scaf, scaf_ndps, scaf_len, scg_off, scg_ids, n_scg, min_ndps=100, max_nodes=1 << 30,
                   separate_fn=None):
    """The work-list recursion of reference src/abawaca.cpp:95-197 around a separate() callable.
    -> (nodes list of dict, scaf2cluster int32[S], dp2cluster int32[N])"""
    separate_fn = separate_fn or separate
    N = keys.shape[1]
    S = len(scaf_ndps)
    work = {1: np.arange(N, dtype=np.int32)}
    next_id = 2
    nodes, scaf2cluster, dp2cluster = [], np.zeros(S, dtype=np.int32), np.zeros(N, dtype=np.int32)
    while work and len(nodes) < max_nodes:
        nid = min(work)
        dps = work.pop(nid)
        res, _per, _raw, side = separate_fn(keys, dp2scaf, scaf_ndps, scaf_len, scg_off, scg_ids, n_scg, dps, min_ndps)
        row = dict(id=nid, ndps=len(dps), split=int(res.split), dim=int(res.dimension), key=int(res.key),
                   spec=float(res.specificity), sens=float(res.sensitivity), c1=0, n1=0, c2=0, n2=0)
        if res.split:
            s_of = side[dp2scaf[dps]]
            c1, c2 = next_id, next_id + 1
            next_id += 2
            work[c1] = dps[s_of == 1]
            work[c2] = dps[s_of == 2]
            row.update(c1=c1, n1=len(work[c1]), c2=c2, n2=len(work[c2]))
        else:
            dp2cluster[dps] = nid
            scaf2cluster[np.unique(dp2scaf[dps])] = nid
        nodes.append(row)
    return nodes, scaf2cluster, dp2cluster


def tinymt_stream(seed: int, n: int):
    st = (C.c_uint32 * 4)()
    lib().abw_or_tinymt_init(st, C.c_uint32(seed))
    return [int(lib().abw_or_tinymt_next(st)) for _ in range(n)]


def kmeans(x, R, base_seed, iters, chunk, nthreads=0):
    """x: fp32 [d, N] dimension-major (same layout as the C ABI)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    d, N = x.shape
    assign = np.empty((R, N), dtype=np.uint8)
    cent = np.empty((R, 2, d), dtype=np.float32)
    inertia = np.empty(R, dtype=np.float64)
    rc = lib().abw_or_kmeans(_p(x, C.c_float), C.c_int64(N), C.c_int32(d), C.c_int32(R), C.c_uint32(base_seed),
                             C.c_int32(iters), C.c_int32(chunk), C.c_int(nthreads),
                             _p(assign, C.c_uint8), _p(cent, C.c_float), _p(inertia, C.c_double))
    if rc != 0:
        raise RuntimeError("abw_or_kmeans failed: %d" % rc)
    return assign, cent, inertia


def kmeans_sharded(x, R, base_seed, iters, chunk, bounds, nthreads=0):
    """abw_or_kmeans_sharded: the points cut into contiguous shards at `bounds` (len nshards + 1)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    d, N = x.shape
    b = np.ascontiguousarray(bounds, dtype=np.int64)
    assign = np.empty((R, N), dtype=np.uint8)
    cent = np.empty((R, 2, d), dtype=np.float32)
    inertia = np.empty(R, dtype=np.float64)
    rc = lib().abw_or_kmeans_sharded(_p(x, C.c_float), C.c_int64(N), C.c_int32(d), C.c_int32(R), C.c_uint32(base_seed),
                                     C.c_int32(iters), C.c_int32(chunk), C.c_int(nthreads), C.c_int32(len(b) - 1),
                                     _p(b, C.c_int64), _p(assign, C.c_uint8), _p(cent, C.c_float), _p(inertia, C.c_double))
    if rc != 0:
        raise RuntimeError("abw_or_kmeans_sharded failed: %d" % rc)
    return assign, cent, inertia


def coverage(seg_scaf, seg_start, seg_end, n_scaf, read_scaf, read_pos, read_len, cov=None):
    """abw_or_coverage: returns cov (fp64 [n_seg]), continuing `cov` when given."""
    seg_scaf = np.ascontiguousarray(seg_scaf, dtype=np.uint32)
    seg_start = np.ascontiguousarray(seg_start, dtype=np.uint64)
    seg_end = np.ascontiguousarray(seg_end, dtype=np.uint64)
    read_scaf = np.ascontiguousarray(read_scaf, dtype=np.uint32)
    read_pos = np.ascontiguousarray(read_pos, dtype=np.uint64)
    read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
    out = np.zeros(len(seg_scaf), dtype=np.float64) if cov is None else np.ascontiguousarray(cov, dtype=np.float64).copy()
    rc = lib().abw_or_coverage(_p(seg_scaf, C.c_uint32), _p(seg_start, C.c_uint64), _p(seg_end, C.c_uint64), C.c_uint64(len(seg_scaf)),
                               C.c_uint64(n_scaf), _p(read_scaf, C.c_uint32), _p(read_pos, C.c_uint64), _p(read_len, C.c_uint32),
                               C.c_uint64(len(read_scaf)), _p(out, C.c_double))
    if rc != 0:
        raise RuntimeError("abw_or_coverage failed: %d" % rc)
    return out


def names_blob(names):
    """scaffold names -> (blob u8, name_off u64[n + 1]) as abw_sam_parse / abw_or_sam_parse take them"""
    enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(e) for e in enc])
    return np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy(), off


def sam_parse(text: np.ndarray, names):
    """abw_or_sam_parse -> (scaf u32, pos0 u64, len u32, flag u32, nsnps u32) of every record, or raises
    ValueError((line number, code)) where the reference would not return a mapping."""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    blob, off = names_blob(names)
    cap = int((text == 10).sum()) + 1
    scaf, pos0 = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint64)
    ln, fl, ns = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
    code = C.c_int32(0)
    n = lib().abw_or_sam_parse(_p(text, C.c_uint8), C.c_uint64(len(text)), _p(blob, C.c_uint8), _p(off, C.c_uint64), C.c_uint64(len(names)),
                               _p(scaf, C.c_uint32), _p(pos0, C.c_uint64), _p(ln, C.c_uint32), _p(fl, C.c_uint32), _p(ns, C.c_uint32),
                               C.c_uint64(cap), C.byref(code))
    if n < 0:
        raise ValueError((int(-n), int(code.value)))
    return scaf[:n], pos0[:n], ln[:n], fl[:n], ns[:n]


def lrn_parse(text: np.ndarray, row_cap: int, ndims_cap: int = 4096):
    """-> (n_rows, ndps_declared, ndims, dp_names int32[n_rows], values f64[ndims, n_rows]); negative n_rows = error"""
    text = np.ascontiguousarray(text, dtype=np.uint8)
    names = np.zeros(max(row_cap, 1), dtype=np.int32)
    vals = np.zeros((ndims_cap, max(row_cap, 1)), dtype=np.float64)
    a, b = C.c_int32(0), C.c_int32(0)
    L = lib()
    L.abw_or_lrn_parse.restype = C.c_int64
    L.abw_or_lrn_parse.argtypes = [C.c_void_p, C.c_uint64, C.c_int64, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p]
    n = L.abw_or_lrn_parse(C.c_void_p(text.ctypes.data), len(text), row_cap, C.byref(a), C.byref(b), C.c_void_p(names.ctypes.data),
                           C.c_void_p(vals.ctypes.data))
    if n < 0:
        return int(n), 0, 0, None, None
    return int(n), a.value, b.value, names[:n], vals[:b.value, :n].copy()


def lrn_emit_rows(values: np.ndarray, dps: np.ndarray) -> np.ndarray:
    """values f64 [n_cols, ld], dps int64 (1-based ids, row order) -> the rows of the cluster's .lrn as bytes"""
    values = np.ascontiguousarray(values, dtype=np.float64)
    dps = np.ascontiguousarray(dps, dtype=np.int64)
    L = lib()
    L.abw_or_lrn_emit_rows.restype = C.c_uint64
    L.abw_or_lrn_emit_rows.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_uint64]
    n_cols, ld = values.shape
    need = L.abw_or_lrn_emit_rows(C.c_void_p(values.ctypes.data), ld, n_cols, C.c_void_p(dps.ctypes.data), len(dps), None, 0)
    text = np.empty(max(int(need), 1), dtype=np.uint8)
    L.abw_or_lrn_emit_rows(C.c_void_p(values.ctypes.data), ld, n_cols, C.c_void_p(dps.ctypes.data), len(dps), C.c_void_p(text.ctypes.data), need)
    return text[:need]
