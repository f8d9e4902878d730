"""The build stage end to end on the GPU (abw_sig_build -> abw_cov_accumulate per SAM file -> abw_lrn_emit, chained by
abawaca_amd/build.py) against the three files the reference's own abawaca-build wrote for the same assembly and SAM
records: abawaca.lrn, abawaca.names and abawaca.info, byte for byte (tests/golden/build_a.npz, made by
tests/golden/make_golden.py from oracle/_ref/abawaca-build: 24 scaffolds with N runs, 3 SAM files with unmapped,
secondary and unknown-reference records and reads hanging over scaffold ends)."""
import os

import numpy as np
import pytest

from abawaca_amd import _lib, build, lrn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _samples(z):
    ro = z["read_off"]
    return [(str(z["sample_names"][s]), z["read_scaf"][ro[s]:ro[s + 1]], z["read_pos1"][ro[s]:ro[s + 1]],
             z["read_len"][ro[s]:ro[s + 1]], z["read_flag"][ro[s]:ro[s + 1]]) +
            ((z["read_nsnps"][ro[s]:ro[s + 1]],) if "read_nsnps" in z else ()) for s in range(len(ro) - 1)]


def test_max_snps_rule_in_the_whole_stage(ctx, golden_dir):
    """build_b: records with 15 and 16 SNPs in their MD:Z / CIGAR strings; .lrn coverage columns and the .info coverage
    (which sums the accepted reads' lengths, ref :242-243) both depend on the rule of ref :546."""
    z = np.load(os.path.join(golden_dir, "build_b.npz"))
    out = build.build_tables(ctx, [str(n) for n in z["scaf_names"]], z["ascii"], z["scaf_off"], _samples(z), this_sample=0)
    assert out["lrn"] == bytes(z["lrn"]) and out["names"] == bytes(z["names"]) and out["info"] == bytes(z["info"])


def test_from_the_text_of_the_sam_files(ctx, golden_dir):
    """The same with the SAM files read on the GPU (abw_sam_parse): FASTA bytes + SAM text in, the reference's three files out."""
    z = np.load(os.path.join(golden_dir, "build_b.npz"))
    samples = [(str(z["sample_names"][s]), z[f"sam{s}"]) for s in range(3)]
    out = build.build_tables(ctx, [str(n) for n in z["scaf_names"]], z["ascii"], z["scaf_off"], samples, this_sample=0)
    assert out["lrn"] == bytes(z["lrn"]) and out["names"] == bytes(z["names"]) and out["info"] == bytes(z["info"])


def test_three_files_byte_identical_to_the_reference(ctx, golden_dir):
    z = np.load(os.path.join(golden_dir, "build_a.npz"))
    out = build.build_tables(ctx, [str(n) for n in z["scaf_names"]], z["ascii"], z["scaf_off"], _samples(z), this_sample=0)
    assert out["lrn"] == bytes(z["lrn"])
    assert out["names"] == bytes(z["names"])
    assert out["info"] == bytes(z["info"])
    # and the GPU loader reads the file it wrote back to the table it was written from
    dp, keys, dims = lrn.lrn_parse(ctx, np.frombuffer(out["lrn"], np.uint8))
    assert np.array_equal(keys.T, out["milli"]) and np.array_equal(dp, np.arange(1, len(dp) + 1))
    assert dims[179:] == [str(n) for n in z["sample_names"]]


def test_scaffold_order_of_the_input_does_not_matter(ctx, golden_dir):
    """The reference keeps its scaffolds in a std::map by name: shuffling the FASTA records gives the same files."""
    z = np.load(os.path.join(golden_dir, "build_a.npz"))
    names = [str(n) for n in z["scaf_names"]]
    off = z["scaf_off"]
    perm = np.random.default_rng(5).permutation(len(names))
    inv = np.empty(len(perm), dtype=np.int64)
    inv[perm] = np.arange(len(perm))
    blob = np.concatenate([z["ascii"][off[i]:off[i + 1]] for i in perm])
    off2 = np.zeros(len(names) + 1, dtype=np.int64)
    off2[1:] = np.cumsum([off[i + 1] - off[i] for i in perm])
    samples = []
    for nm, rs, rp, rl, rf in _samples(z):
        rs2 = np.where(rs < len(names), inv[np.minimum(rs, len(names) - 1)], 0xFFFFFFFF).astype(np.uint32)
        samples.append((nm, rs2, rp, rl, rf))
    out = build.build_tables(ctx, [names[i] for i in perm], blob, off2, samples, this_sample=0)
    assert out["lrn"] == bytes(z["lrn"]) and out["names"] == bytes(z["names"]) and out["info"] == bytes(z["info"])
