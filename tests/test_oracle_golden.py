"""Pins the CPU oracle (oracle/abw_oracle.c) to golden vectors produced by the compiled
reference (tests/golden/make_golden.py).  CPU only."""
import glob
import json
import os

import numpy as np
import pytest

import _oracle as orc


def test_mer_table_matches_reference_header(golden_dir):
    g = np.load(os.path.join(golden_dir, "sig_edge.npz"))
    from abawaca_amd import synth
    mer2dim, names = synth.canonical_mer_table()
    assert np.array_equal(mer2dim, orc.mer_table())
    assert names[0] == "A" and len(names) == 180
    # the reference's '% Key' header row (abawaca-build.cpp:589-593) lists the dims without "A"
    assert list(g["header"]) == names[1:]


@pytest.mark.parametrize("case", ["sig_edge", "sig_rand"])
def test_signature_matches_reference_lrn(golden_dir, case):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    seg_scaf, seg_start, seg_end, dims = orc.build_signatures(g["ascii"], g["scaf_off"])
    # the reference writes rows in scaffold-NAME order; ours are in input order
    names = [str(x) for x in g["scaf_names"]]
    order = sorted(range(len(names)), key=lambda i: names[i].encode())
    idx = np.concatenate([np.flatnonzero(seg_scaf == i) for i in order]) if len(order) else np.zeros(0, int)
    assert np.array_equal(seg_scaf[idx], g["seg_scaf"])
    assert np.array_equal(seg_start[idx], g["seg_start"])
    assert np.array_equal(seg_end[idx], g["seg_end"])
    milli = np.array([orc.milli_of(r) for r in dims[idx]])
    assert np.array_equal(milli[:, 1:], g["milli"].astype(np.int32))   # column 0 is "A", not printed


def test_upcase_rejects_lowercase_n():
    rc, _ = orc.upcase(np.frombuffer(b"ACGTnACGT", dtype=np.uint8))
    assert rc == -1
    rc, s = orc.upcase(np.frombuffer(b"acgtRyN", dtype=np.uint8))
    assert rc == 0 and s.tobytes() == b"ACGTRYN"


def _load_sep(path):
    g = np.load(path)
    S = len(g["scaf_len"])
    scaf_ndps = np.bincount(g["dp2scaf"], minlength=S).astype(np.int32)
    off, ids = orc.scg_csr(g["scg_scaf"], g["scg_id"], S)
    return g, scaf_ndps, off, ids


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sep_*.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_separator_matches_reference(path):
    g, scaf_ndps, off, ids = _load_sep(path)
    nodes, s2c, d2c = orc.recursive_bins(g["milli"], g["dp2scaf"], scaf_ndps, g["scaf_len"], off, ids, int(g["n_scg"]))
    ref = g["nodes"]
    assert len(nodes) == len(ref)
    for mine, r in zip(nodes, ref):
        assert mine["id"] == int(r[0]) and mine["ndps"] == int(r[1])
        assert mine["split"] == int(r[2])
        assert mine["dim"] == int(r[3])
        if mine["dim"] > 0:
            assert mine["key"] / 1000.0 == r[4]       # i/1000.0 == atof("%.3lf") (ClusterData.cpp:159)
        assert mine["spec"] == r[5] and mine["sens"] == r[6]   # bit-exact doubles
        assert (mine["c1"], mine["n1"], mine["c2"], mine["n2"]) == tuple(int(v) for v in r[7:11])
    assert np.array_equal(s2c, g["scaf2cluster"])
    assert np.array_equal(d2c, g["dp2cluster"])


def test_configs0_recursion_matches_reference(golden_dir):
    """BASELINE configs[0] (10 000 scaffolds, 182 columns, whole recursion): the oracle against what the reference's
    separator printed (tests/golden/recursion_c1.npz, made by make_golden.py c1); the table comes from its seed."""
    from abawaca_amd import synth
    g = np.load(os.path.join(golden_dir, "recursion_c1.npz"))
    f = synth.make_features(20240, 10000, 25, 3)
    assert tuple(g["shape"]) == f["milli"].shape and int(g["checksum"][0]) == int(f["milli"].astype(np.int64).sum())
    off, ids = orc.scg_csr(f["scg_scaf"], f["scg_id"], len(f["scaf_len"]))
    nodes, s2c, _d2c = orc.recursive_bins(f["milli"], f["dp2scaf"], f["scaf_ndps"], f["scaf_len"], off, ids, f["n_scg"])
    ref = g["nodes"]
    assert len(nodes) == len(ref) == 49
    for mine, r in zip(nodes, ref):
        assert (mine["id"], mine["ndps"], mine["split"], mine["dim"]) == (int(r[0]), int(r[1]), int(r[2]), int(r[3]))
        assert mine["dim"] <= 0 or mine["key"] / 1000.0 == r[4]
        assert mine["spec"] == r[5] and mine["sens"] == r[6]
        assert (mine["c1"], mine["n1"], mine["c2"], mine["n2"]) == tuple(int(v) for v in r[7:11])
    assert np.array_equal(s2c, g["scaf2cluster"])


def test_tinymt_kat(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "tinymt32_kat.json")))
    for seed, outs in kat["outputs"].items():
        assert orc.tinymt_stream(int(seed), len(outs)) == outs
