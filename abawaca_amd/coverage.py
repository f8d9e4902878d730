"""Row N1 — per-segment read coverage on the GPU (abw_cov_accumulate), host-side glue.

`filter_reads` is the filter of reference src/abawaca-build.cpp:546 -- unmapped (FLAG 0x4), more than max_snps = 15
SNPs (:437), secondary (FLAG 0x100) -- and the 1-based SAM POS -> 0-based position of reference src/ReadMapping.cpp:41.
The SNP count of a mapping is what the reference's ReadMapping::determine_snps collects (src/ReadMapping.cpp:78-185):
`count_snps` restates the count for callers that parse SAM text themselves; a caller that keeps the reference's
SAMReader (integration/abawaca_build_hip.cpp) applies the reference's own filter instead."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import Context, lib


MAX_SNPS = 15      # reference src/abawaca-build.cpp:437


def count_snps(cigar: str, mdz: str) -> int:
    """num_snps() of a mapping (reference src/ReadMapping.cpp:78-185): without an MD:Z tag nothing is collected (:80-81);
    otherwise one SNP per mismatched base and per deleted base of the MD:Z string (:105-121) plus one per 'I' operation
    of the CIGAR string, whatever its length (:158-171).  mdz: the tag's value with or without the "MD:Z:" prefix."""
    if not mdz:
        return 0
    md = mdz[5:] if mdz.startswith("MD:Z:") else mdz
    n, i = 0, 0
    while i < len(md) and md[i].isdigit():
        i += 1
    while i < len(md):
        if md[i] == "^":
            i += 1
            while i < len(md) and not md[i].isdigit():
                n += 1
                i += 1
        else:
            n += 1
            i += 1
        while i < len(md) and md[i].isdigit():
            i += 1
    return n + sum(1 for ch in cigar if ch == "I")


def filter_reads(read_scaf, read_pos1, read_len, read_flag, read_nsnps=None):
    """-> (scaf u32, pos0 u64, len u32) of the mappings the reference accumulates, order kept.  read_nsnps: per mapping
    count_snps(CIGAR, MD:Z); None = no mapping carries an MD:Z tag (then the reference counts no SNPs either)."""
    read_flag = np.asarray(read_flag)
    keep = ((read_flag & 0x4) == 0) & ((read_flag & 0x100) == 0)
    if read_nsnps is not None:
        keep &= np.asarray(read_nsnps) <= MAX_SNPS
    pos0 = np.asarray(read_pos1, dtype=np.int64)[keep] - 1
    return (np.ascontiguousarray(np.asarray(read_scaf, dtype=np.uint32)[keep]),
            np.ascontiguousarray(pos0.astype(np.uint64)),
            np.ascontiguousarray(np.asarray(read_len, dtype=np.uint32)[keep]))


def accumulate(ctx: Context, seg_scaf, seg_start, seg_end, n_scaf: int, read_scaf, read_pos0, read_len, cov=None):
    """cov[g] += overlap / segment length for every read in order (fp64); returns cov."""
    seg_scaf = np.ascontiguousarray(seg_scaf, dtype=np.uint32)
    seg_start = np.ascontiguousarray(seg_start, dtype=np.uint64)
    seg_end = np.ascontiguousarray(seg_end, dtype=np.uint64)
    read_scaf = np.ascontiguousarray(read_scaf, dtype=np.uint32)
    read_pos0 = np.ascontiguousarray(read_pos0, dtype=np.uint64)
    read_len = np.ascontiguousarray(read_len, dtype=np.uint32)
    out = np.zeros(len(seg_scaf), dtype=np.float64) if cov is None else np.ascontiguousarray(cov, dtype=np.float64)
    p = lambda a: C.c_void_p(a.ctypes.data)
    ctx.check(lib().abw_cov_accumulate(ctx.handle, p(seg_scaf), p(seg_start), p(seg_end), len(seg_scaf), int(n_scaf),
                                       p(read_scaf), p(read_pos0), p(read_len), len(read_scaf), p(out)))
    return out


def milli(cov):
    """the thousandths abawaca.lrn prints (ref src/abawaca-build.cpp:603: int(1000.0 * x))"""
    return (1000.0 * np.asarray(cov, dtype=np.float64)).astype(np.int64)


class SamCounts(C.Structure):
    _fields_ = [("lines", C.c_uint64), ("records", C.c_uint64), ("kept", C.c_uint64)]


def names_table(names):
    """scaffold names -> (blob u8, name_off u64[n + 1]) as abw_sam_parse takes them"""
    enc = [n.encode() if isinstance(n, str) else bytes(n) for n in names]
    off = np.zeros(len(enc) + 1, dtype=np.uint64)
    if enc:
        off[1:] = np.cumsum([len(e) for e in enc])
    return np.frombuffer(b"".join(enc) + b"\0", dtype=np.uint8).copy(), off


def sam_names_load(ctx: Context, names) -> int:
    """The name table once per assembly (abw_sam_names_load): afterwards parse_sam(ctx, text, len(names)) -- the NUMBER of
    names in place of the names -- looks RNAME up in the table resident on the device.  Returns len(names)."""
    blob, off = names_table(names)
    ctx.check(lib().abw_sam_names_load(ctx.handle, C.c_void_p(blob.ctypes.data), C.c_void_p(off.ctypes.data), len(names)))
    return len(names)


def parse_sam(ctx: Context, text, names, max_snps: int = MAX_SNPS, every_record: bool = False):
    """SAM text (bytes / uint8 array; a whole file or a part cut at line ends) -> (scaf u32, pos0 u64, len u32) of the records
    the reference accumulates (reference src/ReadMappingReader.cpp:80-116, src/ReadMapping.cpp:23-185,
    src/abawaca-build.cpp:546-551), in file order, ready for `accumulate`.  every_record: also a dict of the arrays of EVERY
    record (scaf, pos0, len, flag, nsnps) and the counts.  Raises (ABW_ERR_FORMAT / ABW_ERR_LOWER_N, the line named in the
    message) where the reference's reader would not return a mapping: fall back to that reader."""
    text = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
    if isinstance(names, int):                                  # the table of sam_names_load(ctx, names): only their number
        n_names, pb, po = names, None, None
    else:
        blob, off = names_table(names)
        n_names, pb, po = len(names), C.c_void_p(blob.ctypes.data), C.c_void_p(off.ctypes.data)
    cap = int(np.count_nonzero(text == 10)) + 1
    rs, rp, rl = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint32)
    p = lambda a: C.c_void_p(a.ctypes.data)
    cnt = SamCounts()
    if every_record:
        cs, cp, cl = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint64), np.zeros(cap, dtype=np.uint32)
        cf, cn = np.zeros(cap, dtype=np.uint32), np.zeros(cap, dtype=np.uint32)
        extra = [p(cs), p(cp), p(cl), p(cf), p(cn)]
    else:
        extra = [None] * 5
    ctx.check(lib().abw_sam_parse(ctx.handle, p(text), len(text), pb, po, n_names, int(max_snps), cap, p(rs), p(rp), p(rl),
                                  *extra, C.byref(cnt)))
    k = int(cnt.kept)
    kept = (rs[:k], rp[:k], rl[:k])
    if not every_record:
        return kept
    r = int(cnt.records)
    return kept, dict(scaf=cs[:r], pos0=cp[:r], len=cl[:r], flag=cf[:r], nsnps=cn[:r], lines=int(cnt.lines), records=r, kept=k)
