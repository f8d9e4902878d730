"""The build stage end to end — host-side mirror of the reference's abawaca-build main loop for the path this repo
replaces (reference src/abawaca-build.cpp:482-490 scaffolds -> segments + signatures, :507-552 SAM records -> coverage
columns, :578-607 abawaca.lrn / abawaca.names / abawaca.info).  The three GPU pieces are abw_sig_build (signature),
abw_cov_accumulate (one call per SAM file) and abw_lrn_emit (the rows of the table); this module only orders, formats
the small per-segment / per-scaffold text and concatenates.  FASTA parsing, the links file (:519-557) and the SCG
prediction (:613-614) stay with the reference's own code: the caller hands over the sequences, and the SAM records either
as arrays or as the text of the file, which then goes through abw_sam_parse (the reference's reader on the GPU).
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

from . import coverage, lrn, signature
from ._lib import Context


def _fmt3(k: int) -> str:
    """"%.3lf" of int(1000.0*x)/1000.0 with k = int(1000.0*x) (ref :597,603)"""
    return "%s%d.%03d" % (("-" if k < 0 else ""), abs(k) // 1000, abs(k) % 1000)


def build_tables(ctx: Context, names: Sequence[str], ascii_blob: np.ndarray, scaf_off: np.ndarray,
                 samples: List[Tuple[str, np.ndarray, np.ndarray, np.ndarray, np.ndarray]], this_sample: int = 0,
                 window: int = 2000):
    """names/ascii_blob/scaf_off: the assembly (any order; the reference keeps a std::map by name, :482-488).
    samples: per SAM file (dimension name = the file's path, :511-512; read_scaf = index into names or 0xFFFFFFFF,
    read_pos1 = SAM POS, read_len, read_flag[, read_nsnps]), in the order of the reference's glob.  read_nsnps =
    coverage.count_snps(CIGAR, MD:Z) per record; this function applies the whole filter of ref :546 (unmapped, more than
    15 SNPs, secondary) before both the coverage sums and the .info coverage.  Without the sixth field no record is taken
    to carry an MD:Z tag.  Or per SAM file (dimension name, the file's text as bytes / uint8 array): read, filtered and
    looked up by coverage.parse_sam.
    -> dict(lrn=bytes, names=bytes, info=bytes, milli=int32[n_dps, 179 + n_samples])"""
    blob = np.ascontiguousarray(ascii_blob, dtype=np.uint8)
    off = np.ascontiguousarray(scaf_off, dtype=np.int64)
    n_scaf = len(off) - 1
    r = signature.build_signatures(ctx, blob, off, window=window, want_dims=False)
    seg_scaf, seg_start, seg_end = r["seg_scaf"].astype(np.int64), r["seg_start"], r["seg_end"]
    n_seg = len(seg_scaf)
    cols = [r["milli"].astype(np.int64)]
    nbps = np.zeros(n_scaf, dtype=np.int64)
    for s, smp in enumerate(samples):
        if len(smp) == 2:
            fs, fp, fl = coverage.parse_sam(ctx, smp[1], names)
        else:
            _dim, rs, rp1, rl, rf = smp[:5]
            fs, fp, fl = coverage.filter_reads(rs, rp1, rl, rf, smp[5] if len(smp) > 5 else None)
        cov = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, n_scaf, fs, fp, fl)
        cols.append(coverage.milli(cov)[:, None])
        if s == this_sample:                                  # Scaf::nbps, ref :242-243: every accepted read of a known scaffold
            known = fs < n_scaf
            nbps += np.bincount(fs[known], weights=fl[known].astype(np.float64), minlength=n_scaf).astype(np.int64)
    milli = np.concatenate(cols, axis=1)
    # rows in the reference's order: scaffolds by name (byte-wise, std::map<string>), segments by position
    name_b = [n.encode() for n in names]
    rank = np.empty(n_scaf, dtype=np.int64)
    order_scaf = sorted(range(n_scaf), key=lambda i: name_b[i])
    rank[order_scaf] = np.arange(n_scaf)
    order = np.lexsort((seg_start, rank[seg_scaf]))
    milli = np.ascontiguousarray(milli[order])
    seg_scaf, seg_start, seg_end = seg_scaf[order], seg_start[order], seg_end[order]
    # upper-cased sequence statistics (ref src/String.cpp:37-51,114-146): N and G/C counts by prefix sums
    up = blob & 0xDF
    cumN = np.concatenate([[0], np.cumsum(up == ord("N"), dtype=np.int64)])
    cumGC = np.concatenate([[0], np.cumsum((up == ord("G")) | (up == ord("C")), dtype=np.int64)])
    # ---- abawaca.lrn (:579-607)
    dim_names = signature.dim_names()[1:] + [s[0] for s in samples]          # all but "A" (:585,591,602)
    head = "%% %d\n%% %d\n%% 9%s\n%% Key%s\n" % (n_seg, len(dim_names) + 1, "\t1" * len(dim_names), "".join("\t" + d for d in dim_names))
    if np.abs(milli).max(initial=0) >= 2 ** 31:
        raise ValueError("a coverage value passes the 32-bit thousandths abw_lrn_emit prints")
    rows = lrn.lrn_emit(ctx, milli.astype(np.int32), 1) if n_seg else np.zeros(0, dtype=np.uint8)
    # ---- abawaca.names (:578,599; the description of ref :219)
    k_in_scaf = np.arange(n_seg) - np.searchsorted(rank[seg_scaf], rank[seg_scaf], side="left") + 1 if n_seg else np.zeros(0, dtype=np.int64)
    a = off[seg_scaf] + seg_start - 1
    b = off[seg_scaf] + seg_end
    size = seg_end - seg_start + 1
    non_n = size - (cumN[b] - cumN[a])
    out_names = ["%% %d\n" % n_seg]
    for i in range(n_seg):
        nm = names[seg_scaf[i]]
        out_names.append("%d\t%s_%d\t%s:(%d, %d), %d/%d non-Ns bps\n" % (i + 1, nm, k_in_scaf[i], nm, seg_start[i], seg_end[i], non_n[i], size[i]))
    # ---- abawaca.info (:597)
    out_info = []
    for i in order_scaf:
        L = int(off[i + 1] - off[i])
        ns = int(cumN[off[i + 1]] - cumN[off[i]])
        gc = 0.0 if L == ns else float(cumGC[off[i + 1]] - cumGC[off[i]]) / float(L - ns)
        cvg = float(nbps[i]) / float(L)
        out_info.append("%s\t%d\t%s\t%s\t%d\n" % (names[i], L, _fmt3(int(1000.0 * cvg)), _fmt3(int(1000.0 * gc)), ns))
    return dict(lrn=head.encode() + rows.tobytes(), names="".join(out_names).encode(), info="".join(out_info).encode(), milli=milli)
