"""abw_sam_parse (row N1, the reader side) against (a) the SAM files a reference run read and the coverage columns it
printed for them (tests/golden/build_b.npz, cov_b.npz), (b) the oracle's restatement of the reference's reader on
generated files with every kind of record, (c) the lines the reference does not survive."""
import os

import numpy as np
import pytest

import _oracle as orc
from _sam_gen import random_sam as _random_sam
from abawaca_amd import _lib, coverage

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _same_as_oracle(ctx, text, names, max_snps=15):
    (rs, rp, rl), ev = coverage.parse_sam(ctx, text, names, max_snps, every_record=True)
    scaf, pos0, ln, fl, ns = orc.sam_parse(text, names)
    assert ev["records"] == len(scaf)
    assert np.array_equal(ev["scaf"], scaf) and np.array_equal(ev["pos0"], pos0) and np.array_equal(ev["len"], ln)
    assert np.array_equal(ev["flag"], fl) and np.array_equal(ev["nsnps"], ns)
    keep = ((fl & 4) == 0) & ((fl & 256) == 0) & (ns <= max_snps)                 # ref src/abawaca-build.cpp:546
    assert np.array_equal(rs, scaf[keep]) and np.array_equal(rp, pos0[keep]) and np.array_equal(rl, ln[keep])
    return ev


def test_the_files_of_the_reference_run_and_its_coverage_columns(ctx):
    g, c = np.load(os.path.join(GOLD, "build_b.npz")), np.load(os.path.join(GOLD, "cov_b.npz"))
    names = [str(n) for n in g["scaf_names"]]
    ro = g["read_off"]
    for s in range(3):
        ev = _same_as_oracle(ctx, g[f"sam{s}"], names)
        a, b = int(ro[s]), int(ro[s + 1])
        assert ev["records"] == b - a and np.array_equal(ev["flag"], g["read_flag"][a:b]) and np.array_equal(ev["nsnps"], g["read_nsnps"][a:b])
        # SAM text -> coverage column, all on this side: what the reference printed
        rs, rp, rl = coverage.parse_sam(ctx, g[f"sam{s}"], names)
        cov = coverage.accumulate(ctx, c["seg_scaf"], c["seg_start"], c["seg_end"], int(c["n_scaf"]), rs, rp, rl)
        assert np.array_equal(coverage.milli(cov), c["cov_milli"][:, s])


@pytest.mark.parametrize("seed,n_lines,n_names,crlf", [(1, 3000, 40, False), (2, 60000, 5000, False), (3, 500, 3, True), (4, 1, 1, False)])
def test_generated_files_against_the_oracle(ctx, seed, n_lines, n_names, crlf):
    names = ["scaffold_%d" % i for i in range(n_names)] + (["scaffold_1"] if n_names > 2 else [])      # a name twice: the first counts
    text = _random_sam(seed, n_lines, names, crlf)
    ev = _same_as_oracle(ctx, text, names)
    assert ev["records"] == n_lines and ev["lines"] == n_lines + 1 + min(5, len(names))
    _same_as_oracle(ctx, text[:-1] if not crlf else text[:-2], names)          # the last line without its end
    _same_as_oracle(ctx, text, names, max_snps=3)
    if n_lines >= 3000:                                                          # the generator does make every kind
        assert ev["nsnps"].max() > 15 and (ev["scaf"] == len(names)).any() and (ev["pos0"] == 2 ** 64 - 1).any()


SEQ, QUAL = "ACGTN" * 10, "I" * 50


@pytest.mark.parametrize("fields,code", [
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ), 1),
    (("r", "x0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL), 2),
    (("r", "0", "s1", "", "255", "50M", "*", "0", "0", SEQ, QUAL), 2),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ[:20] + "n" + SEQ[21:], QUAL), 3),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL, "MD:Z:20a29"), 4),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL, "NM:i:1", "XX:Z:MD:Z:50"), 4),
    (("r", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL[:10], "MD:Z:20A29"), 4),
    (("r", "0", "s1", "5", "255", "*", "*", "0", "0", SEQ, QUAL, "MD:Z:50"), 5),
    (("r", "0", "s1", "5", "255", "5S45M", "*", "0", "0", SEQ, QUAL, "MD:Z:45"), 5),
    (("r", "0", "s1", "5", "255", "49M", "*", "0", "0", SEQ, QUAL, "MD:Z:50"), 6),
    ((), 1),
])
def test_lines_the_reference_does_not_survive(ctx, fields, code):
    good = ("\t".join(["g", "0", "s1", "5", "255", "50M", "*", "0", "0", SEQ, QUAL]) + "\n").encode()
    text = np.frombuffer(good * 3 + ("\t".join(fields) + "\n").encode() + good * 200 + b"also bad\n", dtype=np.uint8)
    with pytest.raises(Exception, match=r"line 4: .*\(code %d\)" % code):
        coverage.parse_sam(ctx, text, ["s0", "s1"])
    with pytest.raises(ValueError) as e:
        orc.sam_parse(text, ["s0", "s1"])
    assert e.value.args[0] == (4, code)


def test_empty_text_and_headers_only(ctx):
    assert [len(a) for a in coverage.parse_sam(ctx, b"", ["s0"])] == [0, 0, 0]
    kept, ev = coverage.parse_sam(ctx, b"@HD\tVN:1.0\n@SQ\tSN:s0\tLN:5\n", ["s0"], every_record=True)
    assert ev["records"] == 0 and ev["lines"] == 2 and len(kept[0]) == 0


def test_mutated_lines_agree_with_the_oracle(ctx):
    """Random edits of one record of a good file (characters that matter to the reader: tabs, digits, bases, CIGAR / MD:Z
    letters, blanks, line ends): the library and the oracle either both read the file to the same arrays or both refuse
    the same line with the same code."""
    rng = np.random.default_rng(99)
    names = ["scaffold_%d" % i for i in range(20)]
    base = bytes(_random_sam(5, 60, names))
    lines = base.split(b"\n")[:-1]
    alphabet = b"\t\t\t0123456789ACGTNMDIZ:^*n =\r-SH"
    refused = agreed = 0
    for trial in range(400):
        ls = list(lines)
        k = int(rng.integers(6, len(ls)))                                  # a record line
        row = bytearray(ls[k])
        for _ in range(int(rng.integers(1, 4))):
            at = int(rng.integers(0, len(row) + 1))
            # edits near the front of the line (the short fields) and in the tags are the interesting ones
            if rng.random() < 0.7:
                tabs = [i for i, c in enumerate(row) if c == 9]
                if len(tabs) >= 10:
                    lo, hi = (0, tabs[5]) if rng.random() < 0.5 else (tabs[9], len(row))
                    at = int(rng.integers(lo, hi + 1))
            op = rng.integers(0, 3)
            ch = alphabet[int(rng.integers(0, len(alphabet)))]
            if op == 0 and at < len(row):
                row[at] = ch
            elif op == 1:
                row.insert(at, ch)
            elif at < len(row):
                del row[at]
        ls[k] = bytes(row)
        text = np.frombuffer(b"\n".join(ls) + b"\n", dtype=np.uint8)
        try:
            want = orc.sam_parse(text, names)
            err = None
        except ValueError as ex:
            err = ex.args[0]
        if err is None:
            (rs, rp, rl), ev = coverage.parse_sam(ctx, text, names, every_record=True)
            assert np.array_equal(ev["scaf"], want[0]) and np.array_equal(ev["pos0"], want[1]) and np.array_equal(ev["len"], want[2])
            assert np.array_equal(ev["flag"], want[3]) and np.array_equal(ev["nsnps"], want[4])
            agreed += 1
        else:
            with pytest.raises(Exception, match=r"line %d: .*\(code %d\)" % err):
                coverage.parse_sam(ctx, text, names)
            refused += 1
    assert refused > 50 and agreed > 50


def test_capacity_is_checked(ctx):
    import ctypes as C
    names = ["scaffold_%d" % i for i in range(10)]
    text = _random_sam(8, 500, names)
    blob, off = coverage.names_table(names)
    cap = 100
    rs, rp, rl = np.zeros(cap, np.uint32), np.zeros(cap, np.uint64), np.zeros(cap, np.uint32)
    cnt = coverage.SamCounts()
    p = lambda a: C.c_void_p(a.ctypes.data)
    rc = _lib.lib().abw_sam_parse(ctx.handle, p(text), len(text), p(blob), p(off), len(names), 15, cap, p(rs), p(rp), p(rl),
                                  None, None, None, None, None, C.byref(cnt))
    assert rc == -3 and cnt.records == 500 and cnt.lines == 506            # ABW_ERR_CAPACITY, counts filled: call again with room
    with pytest.raises(_lib.AbwError):
        ctx.check(rc)


def test_the_resident_name_table_gives_the_same_records():
    """abw_sam_names_load once, then pieces of a file and a second file with names == NULL: the arrays of the calls that
    pass the names; a table of another size, or none, is refused with ABW_ERR_STATE; other work between the pieces (it
    uses the context's scratch buffers) does not disturb the table."""
    c = _lib.Context(0)
    try:
        names = ["scaffold_%d" % i for i in range(700)] + ["scaffold_1"]
        with pytest.raises(_lib.AbwError) as e:
            coverage.parse_sam(c, _random_sam(5, 10, names, False), len(names))
        assert e.value.code == _lib.ABW_ERR_STATE
        n = coverage.sam_names_load(c, names)
        for seed, n_lines in ((6, 4000), (7, 900)):
            text = _random_sam(seed, n_lines, names, False)
            want, ev_w = coverage.parse_sam(c, text, names, every_record=True)
            cut = int(np.flatnonzero(text == 10)[len(names) // 2 + n_lines // 2]) + 1          # a line end in the middle
            got = []
            for piece in (text[:cut], text[cut:]):
                kept, ev = coverage.parse_sam(c, piece, n, every_record=True)
                got.append((kept, ev))
                coverage.parse_sam(c, piece, ["another", "table", "in", "scratch"])           # the per-call table lives elsewhere
            for k in range(3):
                assert np.array_equal(np.concatenate([g[0][k] for g in got]), want[k])
            for key in ("scaf", "pos0", "len", "flag", "nsnps"):
                assert np.array_equal(np.concatenate([g[1][key] for g in got]), ev_w[key])
        with pytest.raises(_lib.AbwError) as e:
            coverage.parse_sam(c, text, n - 1)
        assert e.value.code == _lib.ABW_ERR_STATE
        coverage.sam_names_load(c, [])                                                           # dropped
        with pytest.raises(_lib.AbwError):
            coverage.parse_sam(c, text, n)
    finally:
        c.close()
