"""Row N1, CPU: the coverage restatement (oracle/abw_or_coverage) against the columns the reference's
abawaca-build printed for the same SAM files (tests/golden/cov_a.npz, made by make_golden.py cov)."""
import os

import numpy as np

import _oracle as orc
from abawaca_amd import coverage


def load(golden_dir):
    return np.load(os.path.join(golden_dir, "cov_a.npz"))


def test_oracle_reproduces_reference_coverage_columns(golden_dir):
    g = load(golden_dir)
    n_samples = len(g["read_off"]) - 1
    assert n_samples == 3
    for s in range(n_samples):
        a, b = int(g["read_off"][s]), int(g["read_off"][s + 1])
        rs, rp, rl = coverage.filter_reads(g["read_scaf"][a:b], g["read_pos1"][a:b], g["read_len"][a:b], g["read_flag"][a:b])
        assert len(rs) == (b - a) - 2                           # the unmapped and the secondary line are dropped
        cov = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs, rp, rl)
        assert np.array_equal(coverage.milli(cov), g["cov_milli"][:, s])
        assert cov.min() > 0                                    # every segment saw reads: no column is the reference's
                                                                # out-of-bounds read


def test_chunked_accumulation_continues_the_same_sums(golden_dir):
    g = load(golden_dir)
    a, b = int(g["read_off"][0]), int(g["read_off"][1])
    rs, rp, rl = coverage.filter_reads(g["read_scaf"][a:b], g["read_pos1"][a:b], g["read_len"][a:b], g["read_flag"][a:b])
    whole = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs, rp, rl)
    cut = len(rs) // 3
    part = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs[:cut], rp[:cut], rl[:cut])
    part = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs[cut:], rp[cut:], rl[cut:], cov=part)
    assert np.array_equal(part.view(np.uint64), whole.view(np.uint64))


def test_zero_length_read_follows_the_references_size_t_arithmetic():
    """reference src/abawaca-build.cpp:233 computes e = pos + len - 1 in size_t: with len 0 at pos 0 it wraps and every
    segment of the scaffold receives (end - start + 1) / (end - start + 1) = 1.0; at pos > 0 the segment holding pos - 1
    and pos receives 0 bases.  (Not reachable from a SAM line the reference's parser accepts -- SEQ is never empty --
    so this case has no golden: it restates the arithmetic.)"""
    seg_scaf, seg_start, seg_end = [0, 0, 1, 1, 1], [1, 2001, 1, 1501, 3001], [2000, 4100, 1500, 3000, 4600]
    cov = orc.coverage(seg_scaf, seg_start, seg_end, 2, [1, 0, 1], [0, 2500, 1500], [0, 0, 0])
    assert cov.tolist() == [0.0, 0.0, 1.0, 1.0, 1.0]
