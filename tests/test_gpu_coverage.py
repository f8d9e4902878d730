"""Row N1 on the GPU: abw_cov_accumulate against the reference's printed coverage columns (golden) and,
bit for bit in fp64, against the CPU restatement -- whole files, chunked calls, random read sets with
reads over segment boundaries, scaffold ends, unknown scaffolds and scaffolds without segments."""
import os

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, coverage

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def test_reference_coverage_columns(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "cov_a.npz"))
    for s in range(len(g["read_off"]) - 1):
        a, b = int(g["read_off"][s]), int(g["read_off"][s + 1])
        rs, rp, rl = coverage.filter_reads(g["read_scaf"][a:b], g["read_pos1"][a:b], g["read_len"][a:b], g["read_flag"][a:b])
        cov = coverage.accumulate(ctx, g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs, rp, rl)
        assert np.array_equal(coverage.milli(cov), g["cov_milli"][:, s])
        ref = orc.coverage(g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs, rp, rl)
        assert np.array_equal(cov.view(np.uint64), ref.view(np.uint64))        # same fp64 bits: same order of addition
        # three chunks continue the same sums
        cut1, cut2 = len(rs) // 4, len(rs) // 2
        part = None
        for lo, hi in ((0, cut1), (cut1, cut2), (cut2, len(rs))):
            part = coverage.accumulate(ctx, g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs[lo:hi], rp[lo:hi], rl[lo:hi], cov=part)
        assert np.array_equal(part.view(np.uint64), ref.view(np.uint64))


def test_reference_columns_with_the_max_snps_rule(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "cov_b.npz"))
    for s in range(len(g["read_off"]) - 1):
        a, b = int(g["read_off"][s]), int(g["read_off"][s + 1])
        rs, rp, rl = coverage.filter_reads(g["read_scaf"][a:b], g["read_pos1"][a:b], g["read_len"][a:b], g["read_flag"][a:b], g["read_nsnps"][a:b])
        cov = coverage.accumulate(ctx, g["seg_scaf"], g["seg_start"], g["seg_end"], int(g["n_scaf"]), rs, rp, rl)
        assert np.array_equal(coverage.milli(cov), g["cov_milli"][:, s])


@pytest.mark.parametrize("seed,n_scaf,n_reads", [(1, 50, 20000), (2, 3, 5000), (3, 400, 300000)])
def test_random_reads_bit_exact_vs_oracle(ctx, seed, n_scaf, n_reads):
    rng = np.random.default_rng(seed)
    seg_scaf, seg_start, seg_end, lens = [], [], [], []
    for s in range(n_scaf):
        L = int(rng.integers(500, 12000))
        lens.append(L)
        if rng.random() < 0.1:
            continue                                           # a scaffold without segments
        pos = 1 + int(rng.integers(0, 50))
        while pos + 300 < L:
            ln = int(rng.integers(300, 4000))
            end = min(L, pos + ln - 1)
            seg_scaf.append(s); seg_start.append(pos); seg_end.append(end)
            pos = end + 1 + int(rng.integers(0, 3)) * int(rng.integers(0, 40))   # sometimes a gap (N run cut out)
    lens = np.array(lens)
    rs = rng.integers(0, n_scaf + 2, size=n_reads).astype(np.uint32)            # ids n_scaf, n_scaf+1: unknown names
    rl = rng.integers(1, 300, size=n_reads).astype(np.uint32)
    rp = (rng.random(n_reads) * (lens[np.minimum(rs, n_scaf - 1)] + 100)).astype(np.uint64)   # some hang over the end
    cov = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, n_scaf, rs, rp, rl)
    ref = orc.coverage(seg_scaf, seg_start, seg_end, n_scaf, rs, rp, rl)
    assert np.array_equal(cov.view(np.uint64), ref.view(np.uint64))
    assert ref.max() > 0
    # a coordinate-sorted file: another order of the additions inside every segment, so another oracle run
    order = np.lexsort((rp, rs))
    rs2, rp2, rl2 = rs[order], rp[order], np.full(n_reads, 150, dtype=np.uint32)
    cov2 = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, n_scaf, rs2, rp2, rl2)
    ref2 = orc.coverage(seg_scaf, seg_start, seg_end, n_scaf, rs2, rp2, rl2)
    assert np.array_equal(cov2.view(np.uint64), ref2.view(np.uint64))


@pytest.mark.parametrize("seed,seg_lo,seg_hi,order", [(11, 90, 200, "none"), (12, 90, 200, "coordinate"), (13, 10, 60, "none"),
                                                        (14, 10, 60, "coordinate"), (15, 150, 400, "scaffold"), (16, 150, 400, "coordinate"),
                                                        (17, 100, 130, "coordinate")])
def test_reads_longer_than_segments(ctx, seed, seg_lo, seg_hi, order):
    """A read over three or four segments (coordinate-sorted: the grouped path with K = 3, 4) and over many (pairs), reads
    in no order, coordinate-sorted, and grouped by scaffold only."""
    rng = np.random.default_rng(seed)
    n_scaf, n_reads = 300, 120000
    seg_scaf, seg_start, seg_end, lens = [], [], [], []
    for s in range(n_scaf):
        L = int(rng.integers(400, 3000))
        lens.append(L)
        pos = 1 + int(rng.integers(0, 30))
        while pos + seg_lo < L:
            end = min(L, pos + int(rng.integers(seg_lo, seg_hi)) - 1)
            seg_scaf.append(s); seg_start.append(pos); seg_end.append(end)
            pos = end + 1 + (int(rng.integers(0, 25)) if rng.random() < 0.2 else 0)
    lens = np.array(lens)
    rs = rng.integers(0, n_scaf + 1, size=n_reads).astype(np.uint32)             # id n_scaf: an unknown name
    rl = rng.integers(1, 300, size=n_reads).astype(np.uint32)
    rp = (rng.random(n_reads) * (lens[np.minimum(rs, n_scaf - 1)] + 50)).astype(np.uint64)
    if order == "coordinate":
        o = np.lexsort((rp, rs)); rs, rp, rl = rs[o], rp[o], rl[o]
    elif order == "scaffold":
        o = np.argsort(rs, kind="stable"); rs, rp, rl = rs[o], rp[o], rl[o]
    cov = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, n_scaf, rs, rp, rl)
    ref = orc.coverage(seg_scaf, seg_start, seg_end, n_scaf, rs, rp, rl)
    assert np.array_equal(cov.view(np.uint64), ref.view(np.uint64))
    # a second sample onto the same accumulators
    cov2 = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, n_scaf, rs[::-1].copy(), rp[::-1].copy(), rl[::-1].copy(), cov=cov.copy())
    ref2 = orc.coverage(seg_scaf, seg_start, seg_end, n_scaf, rs[::-1].copy(), rp[::-1].copy(), rl[::-1].copy(), cov=ref.copy())
    assert np.array_equal(cov2.view(np.uint64), ref2.view(np.uint64))


def test_zero_length_reads_as_the_reference_computes_them(ctx):
    seg_scaf, seg_start, seg_end = [0, 0, 1, 1, 1], [1, 2001, 1, 1501, 3001], [2000, 4100, 1500, 3000, 4600]
    rs, rp, rl = [1, 0, 1, 0, 1], [0, 2500, 1500, 100, 0], [0, 0, 0, 150, 7]
    cov = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, 2, rs, rp, rl)
    ref = orc.coverage(seg_scaf, seg_start, seg_end, 2, rs, rp, rl)
    assert np.array_equal(cov.view(np.uint64), ref.view(np.uint64)) and cov[3] == 1.0 and cov[1] == 0.0


def test_a_segment_of_2_to_the_32_bases_is_refused(ctx):
    with pytest.raises(Exception, match="2\\^32 bases"):
        coverage.accumulate(ctx, [0, 0], [1, 1 << 33], [1 << 32, 1 << 34], 1, [0], [5], [100])


def test_empty_inputs(ctx):
    z = np.zeros(0)
    assert coverage.accumulate(ctx, z, z, z, 0, z, z, z).shape == (0,)
    cov = coverage.accumulate(ctx, [0], [1], [100], 1, z, z, z)
    assert cov.tolist() == [0.0]


def test_bench_sized_read_sets_bit_exact_vs_oracle(ctx):
    """4 M reads of 150 bases on 200 000 segments (the bench's shape, a fifth of its reads): in no order (pairs through the
    radix sort, 18-bit keys, many tiles) and coordinate-sorted (grouped reads summed where they lie)."""
    rng = np.random.default_rng(77)
    S = 100_000
    seg_scaf = np.repeat(np.arange(S, dtype=np.uint32), 2)
    seg_start = np.tile(np.array([1, 5001], dtype=np.uint64), S)
    seg_end = np.tile(np.array([5000, 10000], dtype=np.uint64), S)
    n = 4_000_000
    rs = rng.integers(0, S, size=n).astype(np.uint32)
    rp = rng.integers(0, 10_000 - 150, size=n).astype(np.uint64)
    rl = np.full(n, 150, dtype=np.uint32)
    cov = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, S, rs, rp, rl)
    ref = orc.coverage(seg_scaf, seg_start, seg_end, S, rs, rp, rl)
    assert np.array_equal(cov.view(np.uint64), ref.view(np.uint64))
    o = np.lexsort((rp, rs))
    cov2 = coverage.accumulate(ctx, seg_scaf, seg_start, seg_end, S, rs[o], rp[o], rl[o])
    ref2 = orc.coverage(seg_scaf, seg_start, seg_end, S, rs[o], rp[o], rl[o])
    assert np.array_equal(cov2.view(np.uint64), ref2.view(np.uint64))
    assert not np.array_equal(cov.view(np.uint64), cov2.view(np.uint64))       # the order of the additions is part of the result
