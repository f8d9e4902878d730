"""The oracle's SAM reader (abw_or_sam_parse: Bio::ReadMapping's constructor and determine_snps restated) pinned to the
SAM files the reference's own abawaca-build read for tests/golden/build_b.npz: the arrays kept there, put through the
filter of src/abawaca-build.cpp:546, give the coverage columns the reference printed (tests/test_coverage_oracle.py), so
a reader that reproduces the arrays reproduces what the reference made of the files."""
import os

import numpy as np
import pytest

import _oracle as orc

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _line(*fields):
    return np.frombuffer(("\t".join(fields) + "\n").encode(), dtype=np.uint8)


def test_reader_reproduces_the_arrays_of_the_reference_run():
    g = np.load(os.path.join(GOLD, "build_b.npz"))
    names = [str(n) for n in g["scaf_names"]]
    ro = g["read_off"]
    for s in range(3):
        scaf, pos0, ln, fl, ns = orc.sam_parse(g[f"sam{s}"], names)
        a, b = int(ro[s]), int(ro[s + 1])
        assert len(scaf) == b - a
        want_scaf = g["read_scaf"][a:b].astype(np.int64)
        want_scaf[want_scaf == 0xFFFFFFFF] = len(names)              # the fixture marks unknown names with all ones
        assert np.array_equal(scaf, want_scaf)
        assert np.array_equal(pos0, (g["read_pos1"][a:b] - 1).astype(np.uint64))      # POS 0 wraps, as size_t does
        assert np.array_equal(ln, g["read_len"][a:b]) and np.array_equal(fl, g["read_flag"][a:b])
        assert np.array_equal(ns, g["read_nsnps"][a:b])
        assert ns.max() == 16 and (fl & 4).any() and (fl & 256).any() and (scaf == len(names)).any()


SEQ, QUAL = "ACGTN" * 10, "I" * 50


@pytest.mark.parametrize("fields,code", [
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ), 1),                              # ten fields
    (("r", "x0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL), 2),
    (("r", "0", "s1", "-5", "255", "50M", "*", "0", "0", SEQ, QUAL), 2),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ[:20] + "n" + SEQ[21:], QUAL), 3),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL, "MD:Z:20a29"), 4),             # a lower-case reference base
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL, "MD:Z:20A"), 4),               # no number behind the base
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL, "XX:Z:MD:Z:50"), 4),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL[:10], "MD:Z:20A29"), 4),        # QUAL shorter than the SNP's place
    (("r", "0", "s1", "5", "255", "*", "*", "0", "0", SEQ, QUAL, "MD:Z:50"), 5),
    (("r", "0", "s1", "5", "255", "5S45M", "*", "0", "0", SEQ, QUAL, "MD:Z:45"), 5),
    (("r", "0", "s1", "5", "255", "M50", "*", "0", "0", SEQ, QUAL, "MD:Z:50"), 5),
    (("r", "0", "s1", "5", "255", "49M", "*", "0", "0", SEQ, QUAL, "MD:Z:50"), 6),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL, "MD:Z:49"), 6),
    ((), 1),                                                                                      # an empty line
])
def test_lines_the_reference_does_not_survive(fields, code):
    good = _line("g", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL)
    with pytest.raises(ValueError) as e:
        orc.sam_parse(np.concatenate([good, _line(*fields), good]), ["s0", "s1"])
    assert e.value.args[0] == (2, code)


def test_counts_and_quirks():
    text = np.concatenate([
        np.frombuffer(b"@HD\tVN:1.0\n@SQ\tSN:s1\tLN:1000\n", dtype=np.uint8),
        _line("a", "0", "s1", "0", "255", "50M", "*", "0", "0", SEQ, QUAL),                       # POS 0: pos0 wraps
        _line("b", "16", "s0", "7", "255", "20M2I28M", "=", "0", "0", SEQ, QUAL, "NM:i:3", "MD:Z:10A37"),
        _line("c", "0", "s9", "7", "255", "25M3D25M", "*", "0", "0", SEQ, QUAL, "MD:Z:25^ACG0T24"),
        _line("d", "4", "*", "0", "0", "*", "*", "0", "0", "*", "*"),                             # SEQ "*": one base long
        np.frombuffer(b"e\t0\ts1\t9\t255\t50M\t*\t0\t0\t" + SEQ.encode() + b"\t" + QUAL.encode() + b"\r\n", dtype=np.uint8),
        np.frombuffer(b"f\t0\ts1\t9\t255\t50M\t*\t0\t0\t" + SEQ.encode() + b"\t" + QUAL.encode(), dtype=np.uint8),      # no end of line
    ])
    scaf, pos0, ln, fl, ns = orc.sam_parse(text, ["s0", "s1"])
    assert scaf.tolist() == [1, 0, 2, 2, 1, 1] and fl.tolist() == [0, 16, 0, 4, 0, 0]
    assert pos0.tolist() == [2 ** 64 - 1, 6, 6, 2 ** 64 - 1, 8, 8] and ln.tolist() == [50, 50, 50, 1, 50, 50]
    assert ns.tolist() == [0, 2, 4, 0, 0, 0]


def test_reader_against_the_reference_reader_itself(tmp_path):
    """oracle/_ref/ref_sam_reader = the reference's own SAMReader / ReadMapping around a counting loop (compiled from the
    reference's sources): records, records kept by the filter and the sums of the fields over a generated file."""
    import subprocess
    from _sam_gen import random_sam
    exe = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "ref_sam_reader")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_sam_reader is built where /root/reference exists (make -C oracle ref)")
    names = ["scaffold_%d" % i for i in range(300)]
    for seed, crlf in ((21, False), (22, True)):
        text = random_sam(seed, 20000, names, crlf)
        p = tmp_path / f"s{seed}.sam"
        p.write_bytes(text.tobytes())
        out = subprocess.run([exe, str(p)], capture_output=True, text=True, timeout=120, check=True).stdout.split()
        scaf, pos0, ln, fl, ns = orc.sam_parse(text, names)
        keep = ((fl & 4) == 0) & ((fl & 256) == 0) & (ns <= 15)
        want = [len(scaf), int(keep.sum()), int((fl & 0x104).sum()), int(pos0.sum(dtype=np.uint64)), int(ln.sum()), int(ns.sum())]
        assert [int(x) for x in out[:6]] == want


def test_mutated_lines_against_the_reference_reader(tmp_path):
    """Random edits of one record of a good file.  What the oracle reads, the reference's own reader reads to the same sums;
    what the oracle refuses because the reference throws (codes 3 .. 6), the reference's reader does not survive (it ends on
    the uncaught exception).  Codes 1 (the reference indexes a field that is not there: undefined) and 2 (atoi would stop
    at the stray character: the strictness of this restatement) are not the reference's to confirm."""
    import subprocess
    from _sam_gen import random_sam
    exe = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "ref_sam_reader")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/ref_sam_reader is built where /root/reference exists (make -C oracle ref)")
    rng = np.random.default_rng(123)
    names = ["scaffold_%d" % i for i in range(20)]
    lines = bytes(random_sam(6, 60, names)).split(b"\n")[:-1]
    alphabet = b"0123456789ACGTNMDIZ:^*n =-SH"
    read = thrown = 0
    p = tmp_path / "m.sam"
    for trial in range(250):
        ls = list(lines)
        k = int(rng.integers(6, len(ls)))
        row = bytearray(ls[k])
        tabs = [i for i, c in enumerate(row) if c == 9]
        for _ in range(int(rng.integers(1, 3))):
            lo, hi = (tabs[4] + 1, tabs[5]) if rng.random() < 0.4 else (tabs[9] + 1, len(row))     # CIGAR, or SEQ .. the tags
            at = int(rng.integers(lo, max(lo + 1, hi)))
            ch = alphabet[int(rng.integers(0, len(alphabet)))]
            if rng.random() < 0.6 and at < len(row):
                row[at] = ch
            else:
                row.insert(at, ch)
            tabs = [i for i, c in enumerate(row) if c == 9]
        ls[k] = bytes(row)
        text = np.frombuffer(b"\n".join(ls) + b"\n", dtype=np.uint8)
        p.write_bytes(text.tobytes())
        out = subprocess.run([exe, str(p)], capture_output=True, text=True, timeout=60)
        try:
            scaf, pos0, ln, fl, ns = orc.sam_parse(text, names)
        except ValueError as ex:
            line, code = ex.args[0]
            if code >= 3:
                assert out.returncode != 0, (code, ls[k])
                thrown += 1
            continue
        assert out.returncode == 0, (out.stderr[-300:], ls[k])
        keep = ((fl & 4) == 0) & ((fl & 256) == 0) & (ns <= 15)
        want = [len(scaf), int(keep.sum()), int((fl & 0x104).sum()), int(pos0.sum(dtype=np.uint64)), int(ln.sum()), int(ns.sum())]
        assert [int(x) for x in out.stdout.split()[:6]] == want, ls[k]
        read += 1
    assert read > 30 and thrown > 30
