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


def test_max_snps_rule_on_md_tags(golden_dir):
    """tests/golden/cov_b.npz: SAM files whose records carry MD:Z / CIGAR strings with 15 and 16 SNPs (mismatches, deleted
    bases, insertions) -- the reference (src/abawaca-build.cpp:546, src/ReadMapping.cpp:78-185) keeps the former and drops
    the latter; coverage.count_snps + filter_reads must do the same for the printed columns to come out."""
    g = np.load(os.path.join(golden_dir, "cov_b.npz"))
    assert coverage.count_snps("150M", "MD:Z:" + "8A" * 15 + "15") == 15 and coverage.count_snps("150M", "") == 0
    assert coverage.count_snps("60M1I10M1I78M", "MD:Z:" + "8A" * 14 + "22") == 16
    assert coverage.count_snps("75M4D75M", "75^ACGT" + "5A" * 12 + "3") == 16
    changed = 0
    for s in range(len(g["read_off"]) - 1):
        a, b = int(g["read_off"][s]), int(g["read_off"][s + 1])
        ns = g["read_nsnps"][a:b]
        assert sorted(set(ns.tolist())) == [0, 1, 15, 16]
        rs, rp, rl = coverage.filter_reads(g["read_scaf"][a:b], g["read_pos1"][a:b], g["read_len"][a:b], g["read_flag"][a:b], ns)
        cov = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs, rp, rl)
        assert np.array_equal(coverage.milli(cov), g["cov_milli"][:, s])
        rs0, rp0, rl0 = coverage.filter_reads(g["read_scaf"][a:b], g["read_pos1"][a:b], g["read_len"][a:b], g["read_flag"][a:b])
        assert len(rs0) == len(rs) + 9                           # three places x three records with 16 SNPs
        cov0 = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs0, rp0, rl0)
        changed += int((coverage.milli(cov0) != g["cov_milli"][:, s]).sum())
    assert changed > 0                                           # without the rule the printed columns differ
