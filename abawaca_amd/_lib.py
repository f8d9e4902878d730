"""ctypes binding of libabawaca_hip.so (the C ABI declared in include/abawaca_hip.h).

There is NO fallback: if the shared library is missing or does not load, importing this module's
`lib()` raises.  The oracle under oracle/ is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABW_LIB_PATH") or os.path.join(_HERE, "libabawaca_hip.so")   # override: kernel experiments only

ABW_OK = 0
ABW_ERR_ARG, ABW_ERR_HIP, ABW_ERR_CAPACITY, ABW_ERR_LOWER_N, ABW_ERR_STATE, ABW_ERR_NOMEM = -1, -2, -3, -4, -5, -6
ABW_ERR_FORMAT = -7
SIG_NDIMS, SIG_NCOLS, SIG_NMERS = 180, 179, 340

# every symbol include/abawaca_hip.h declares (tests/test_abi.py checks the two lists agree)
EXPORTS = [
    "abw_ctx_create", "abw_ctx_destroy", "abw_ctx_set_stream", "abw_ctx_synchronize", "abw_last_error",
    "abw_abi_version", "abw_prof_enable", "abw_prof_reset", "abw_prof_read", "abw_sig_dim_names", "abw_sig_segment_host", "abw_sig_segment_dev", "abw_sig_segment_static", "abw_sig_count_dev", "abw_sig_selftest_division",
    "abw_sig_build", "abw_sep_load", "abw_sep_load_dev", "abw_km_sum_ordered", "abw_sam_parse", "abw_sam_parse_dev", "abw_sam_names_load", "abw_sep_node", "abw_sep_node_assign", "abw_sep_tree", "abw_sep_dev_keys", "abw_km_run_dev", "abw_km_run",
    "abw_km_features_from_keys", "abw_km_tinymt_selftest", "abw_km_auto_chunk", "abw_km_chunk_for", "abw_km_set_arith",
    "abw_cov_accumulate", "abw_cov_accumulate_dev", "abw_km_load", "abw_km_node", "abw_lrn_header", "abw_lrn_parse", "abw_lrn_emit", "abw_lrn_emit_rows", "abw_lrn_emit_rows_dev", "abw_lrn_table_load", "abw_lrn_table_load_dev",
    "abw_sep_shard_load", "abw_sep_shard_accum", "abw_sep_shard_eval", "abw_sep_shard_decide", "abw_sep_shard_combine", "abw_device_count",
    "abw_sep_tree_begin", "abw_sep_tree_eval", "abw_sep_tree_assign", "abw_sep_tree_advance", "abw_sep_tree_result", "abw_sep_tree_merge",
    "abw_sep_tree_level", "abw_sep_narrow_bins", "abw_sep_stree_begin", "abw_sep_stree_accum", "abw_sep_stree_eval", "abw_sep_stree_assign",
    "abw_km_shard_begin", "abw_km_shard_ready", "abw_km_shard_step", "abw_km_shard_update", "abw_km_shard_finish",
]


class AbwError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"abawaca_hip error {code}: {msg}")
        self.code = code


class DimBest(C.Structure):
    _fields_ = [("found", C.c_int32), ("key", C.c_int32), ("sensitivity", C.c_double), ("specificity", C.c_double)]


class KmSplit(C.Structure):
    _fields_ = [("split", C.c_int32), ("best_restart", C.c_int32), ("inertia", C.c_double), ("agreement", C.c_double),
                ("ndps", C.c_int64), ("raw1", C.c_int64), ("ndps1", C.c_int64), ("ndps2", C.c_int64),
                ("chunk", C.c_int32), ("reserved", C.c_int32)]


class TreeNode(C.Structure):
    _fields_ = [("id", C.c_int32), ("n_scaf", C.c_int32), ("ndps", C.c_int64), ("split", C.c_int32), ("dimension", C.c_int32),
                ("key", C.c_int32), ("legal", C.c_int32), ("sensitivity", C.c_double), ("specificity", C.c_double),
                ("child1", C.c_int32), ("child2", C.c_int32), ("ndps1", C.c_int64), ("ndps2", C.c_int64)]


class Split(C.Structure):
    _fields_ = [("split", C.c_int32), ("dimension", C.c_int32), ("key", C.c_int32), ("legal", C.c_int32),
                ("sensitivity", C.c_double), ("specificity", C.c_double), ("ndps1", C.c_int64), ("ndps2", C.c_int64)]


_LIB = None


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `make -C abawaca_amd/csrc` (or __graft_entry__.build()). "
            "abawaca_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    vp, u64, i64, i32, u32 = C.c_void_p, C.c_uint64, C.c_int64, C.c_int32, C.c_uint32
    L.abw_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.abw_ctx_destroy.argtypes = [vp]
    L.abw_ctx_destroy.restype = None
    L.abw_ctx_set_stream.argtypes = [vp, vp]
    L.abw_ctx_synchronize.argtypes = [vp]
    L.abw_last_error.argtypes = [vp]
    L.abw_last_error.restype = C.c_char_p
    L.abw_prof_enable.argtypes = [vp, C.c_int]
    L.abw_prof_reset.argtypes = [vp]
    L.abw_prof_read.argtypes = [vp, C.c_char_p, C.POINTER(u64), C.POINTER(C.c_double)]
    L.abw_sig_dim_names.argtypes = [C.c_char_p, u64]
    L.abw_sig_segment_host.argtypes = [vp, vp, u64, u32, u64, vp, vp, vp, C.POINTER(u64)]
    L.abw_sig_segment_dev.argtypes = [vp, vp, u64, vp, u64, u32, u64, vp, vp, vp, C.POINTER(u64)]
    L.abw_sig_segment_static.argtypes = [vp, vp, u64, vp, u64, u32, u64, vp, vp, vp, vp]
    L.abw_sig_selftest_division.argtypes = [vp, u32, u32, u32, u32, C.POINTER(u64), vp]
    L.abw_sig_count_dev.argtypes = [vp, vp, u64, vp, vp, vp, vp, u64, vp, vp, vp]
    L.abw_sig_build.argtypes = [vp, vp, vp, u64, u32, u64, vp, vp, vp, vp, vp, C.POINTER(u64)]
    L.abw_sep_load.argtypes = [vp, vp, i32, i64, vp, vp, i32, vp, vp, i32]
    L.abw_sam_parse.argtypes = [vp, vp, u64, vp, vp, u64, u32, u64] + [vp] * 9
    L.abw_sam_names_load.argtypes = [vp, vp, vp, u64]
    L.abw_sam_parse_dev.argtypes = [vp, vp, u64, vp, vp, u64, u32, u64] + [vp] * 9
    L.abw_km_sum_ordered.argtypes = [vp, vp, i32, i64, vp]
    L.abw_sep_load_dev.argtypes = [vp, vp, i32, i64, vp, vp, i32, vp, vp, i32]
    L.abw_sep_node.argtypes = [vp, vp, i32, i32, C.POINTER(Split), vp, vp, vp]
    L.abw_sep_node_assign.argtypes = [vp, vp, i32, i32, i32, i32, C.POINTER(Split), vp, vp]
    L.abw_sep_tree.argtypes = [vp, i32, i64, vp, i64, C.POINTER(i64), vp]
    L.abw_sep_dev_keys.argtypes = [vp]
    L.abw_sep_shard_load.argtypes = [vp, vp, i32, i64, vp, vp, i32, vp, vp, i32, vp, vp]
    L.abw_sep_shard_accum.argtypes = [vp, vp, i32, vp]
    L.abw_sep_shard_eval.argtypes = [vp, i32, C.POINTER(Split), vp, vp, vp]
    L.abw_sep_shard_decide.argtypes = [C.POINTER(Split), i64, i64, i32]
    L.abw_sep_shard_combine.argtypes = [C.POINTER(vp), i32]
    L.abw_device_count.argtypes = [C.POINTER(i32)]
    L.abw_sep_tree_begin.argtypes = [vp, i32, i64]
    L.abw_sep_tree_eval.argtypes = [vp, vp, i64, C.POINTER(i64)]
    L.abw_sep_tree_assign.argtypes = [vp, vp, i64, C.POINTER(vp), C.POINTER(i64)]
    L.abw_sep_tree_advance.argtypes = [vp]
    L.abw_sep_tree_result.argtypes = [vp, vp, i64, C.POINTER(i64), vp]
    L.abw_sep_tree_merge.argtypes = [vp, i32, i64, vp, vp]
    L.abw_sep_tree_level.argtypes = [vp, vp, vp, i64]
    L.abw_sep_narrow_bins.argtypes = [vp, vp, i32]
    L.abw_sep_stree_begin.argtypes = [vp, i32]
    L.abw_sep_stree_accum.argtypes = [vp, vp, i64, vp, vp]
    L.abw_sep_stree_eval.argtypes = [vp, i32, vp, i64]
    L.abw_sep_stree_assign.argtypes = [vp, vp, i64, vp, vp]
    L.abw_sep_dev_keys.restype = vp
    L.abw_km_run_dev.argtypes = [vp, vp, i64, i32, i32, u32, i32, i32, vp, vp, vp]
    L.abw_km_run.argtypes = [vp, vp, i64, i32, i32, u32, i32, i32, vp, vp, vp]
    L.abw_km_features_from_keys.argtypes = [vp, vp, i32, i64, vp, vp]
    L.abw_km_tinymt_selftest.argtypes = [vp, u32, i32, vp]
    L.abw_cov_accumulate.argtypes = [vp, vp, vp, vp, u64, u64, vp, vp, vp, u64, vp]
    L.abw_cov_accumulate_dev.argtypes = [vp, vp, vp, vp, u64, u64, vp, vp, vp, u64, vp]
    L.abw_km_shard_begin.argtypes = [vp, vp, i64, i64, i64, i32, i32, u32, i32, vp]
    L.abw_km_shard_ready.argtypes = [vp]
    L.abw_km_shard_step.argtypes = [vp, vp]
    L.abw_km_shard_update.argtypes = [vp, i32]
    L.abw_km_shard_finish.argtypes = [vp, vp, vp]
    L.abw_lrn_header.argtypes = [vp, u64, C.POINTER(i64), C.POINTER(i32), C.c_char_p, u64]
    L.abw_lrn_parse.argtypes = [vp, vp, u64, i64, i32, vp, vp]
    L.abw_lrn_emit.argtypes = [vp, vp, i64, i32, u64, vp, u64, C.POINTER(u64)]
    L.abw_lrn_emit_rows.argtypes = [vp, vp, i64, i32, vp, i64, vp, u64, C.POINTER(u64)]
    L.abw_lrn_table_load.argtypes = [vp, vp, i64, i32]
    L.abw_lrn_table_load_dev.argtypes = [vp, vp, i64, i32]
    L.abw_lrn_emit_rows_dev.argtypes = [vp, vp, i64, i32, vp, i64, vp, u64, C.POINTER(u64), vp]
    L.abw_km_load.argtypes = [vp, vp, i32, i64, vp, i32]
    L.abw_km_node.argtypes = [vp, vp, i32, i32, u32, i32, i32, i32, C.POINTER(KmSplit), vp, vp]
    L.abw_km_auto_chunk.argtypes = [vp, i64, i32]
    L.abw_km_auto_chunk.restype = i32
    L.abw_km_chunk_for.argtypes = [i32, i64, i32]
    L.abw_km_chunk_for.restype = i32
    L.abw_km_set_arith.argtypes = [vp, i32]
    _LIB = L
    return L


class Context:
    """Owns one abw_ctx (one GPU, one stream)."""

    def __init__(self, device: int = 0):
        self._h = C.c_void_p()
        rc = lib().abw_ctx_create(int(device), C.byref(self._h))
        if rc != ABW_OK:
            raise AbwError(rc, "abw_ctx_create failed (is a GPU visible?)")
        self.device = int(device)

    def check(self, rc):
        if rc != ABW_OK:
            raise AbwError(rc, lib().abw_last_error(self._h).decode(errors="replace"))

    @property
    def handle(self):
        return self._h

    def set_stream(self, hip_stream_ptr):
        self.check(lib().abw_ctx_set_stream(self._h, C.c_void_p(hip_stream_ptr)))

    shares_torch_stream = False

    def use_torch_stream(self, on=True):
        """Run the library on torch's current stream of this device (or back on its own): work that alternates between
        the library and torch / RCCL is then ordered on the device and needs no host synchronisation in between."""
        import torch
        if on:
            self.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        else:
            self.set_stream(0)
        self.shares_torch_stream = bool(on)

    def prof_enable(self, on=True):
        """on: False / True (every launch bracketed by events) or an int n > 1 (every n-th launch of a family: a sample)."""
        self.check(lib().abw_prof_enable(self._h, int(on)))

    def prof_reset(self):
        self.check(lib().abw_prof_reset(self._h))

    def prof_read(self, name: str):
        n, ms = C.c_uint64(0), C.c_double(0.0)
        self.check(lib().abw_prof_read(self._h, name.encode(), C.byref(n), C.byref(ms)))
        return int(n.value), float(ms.value)

    def synchronize(self):
        self.check(lib().abw_ctx_synchronize(self._h))

    def close(self):
        if self._h:
            lib().abw_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
