"""abw_sep_tree -- the whole recursion of reference src/abawaca.cpp:95-197 in one call, level by level on the device --
against (a) what the reference's own separator printed (goldens: six small assemblies and BASELINE configs[0]) and
(b) abw_sep_node called node by node in work-list order (separator.binning), which the other tests pin to the oracle:
the same node ids, winners (dimension, key, sensitivity / specificity bits), children, sizes and final bins."""
import glob
import os

import numpy as np
import pytest

from abawaca_amd import _lib, separator, synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sep_*.npz")))


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _key(r):
    return (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])


def _same_as_node_by_node(store, max_nodes=0):
    want, s2c, _d2c = separator.binning(store, max_nodes=max_nodes or (1 << 30))
    got, s2c_t, _dt = separator.binning_tree(store, max_nodes=max_nodes)
    assert [_key(r) for r in got] == [_key(r) for r in want]
    assert np.array_equal(s2c_t, s2c)
    return got


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_tree_matches_reference_goldens(ctx, path):
    g = np.load(path)
    S = len(g["scaf_len"])
    off, ids = separator.scg_csr(g["scg_scaf"], g["scg_id"], S)
    store = separator.ColumnStore(ctx, g["milli"], g["dp2scaf"], g["scaf_len"], off, ids, int(g["n_scg"]))
    got, s2c, _dt = separator.binning_tree(store)
    ref = g["nodes"]
    assert len(got) == len(ref)
    for a, r in zip(got, ref):
        assert (a["id"], a["ndps"], a["split"], a["dim"]) == (int(r[0]), int(r[1]), int(r[2]), int(r[3]))
        assert a["dim"] <= 0 or a["key"] / 1000.0 == r[4]
        assert a["spec"] == r[5] and a["sens"] == r[6]
        assert (a["c1"], a["n1"], a["c2"], a["n2"]) == tuple(int(v) for v in r[7:11])
    assert np.array_equal(s2c, g["scaf2cluster"])
    _same_as_node_by_node(store)
    for cut in (1, 2, 4):                                         # a work list cut after some nodes
        _same_as_node_by_node(store, max_nodes=cut)


def test_tree_on_configs0_matches_the_reference(ctx, golden_dir):
    g = np.load(os.path.join(golden_dir, "recursion_c1.npz"))
    f = synth.make_features(20240, 10000, 25, 3)
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], len(f["scaf_len"]))
    store = separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    got, s2c, _dt = separator.binning_tree(store)
    ref = g["nodes"]
    assert len(got) == len(ref) == 49
    for a, r in zip(got, ref):
        assert (a["id"], a["ndps"], a["split"], a["dim"]) == (int(r[0]), int(r[1]), int(r[2]), int(r[3]))
        assert a["dim"] <= 0 or a["key"] / 1000.0 == r[4]
        assert a["spec"] == r[5] and a["sens"] == r[6]
        assert (a["c1"], a["n1"], a["c2"], a["n2"]) == tuple(int(v) for v in r[7:11])
    assert np.array_equal(s2c, g["scaf2cluster"])


@pytest.mark.parametrize("seed,n_scaf,G,S_cov,n_hi", [(1, 3000, 6, 3, 14), (2, 12000, 12, 10, 6), (3, 30000, 30, 4, 40)])
def test_tree_matches_node_by_node_on_synthetic_tables(ctx, seed, n_scaf, G, S_cov, n_hi):
    """incl. scaffolds with more than 8 dps (the other order-statistic path) and wide coverage columns"""
    f = synth.make_features(seed, n_scaf, G, S_cov, n_lo=2, n_hi=n_hi)
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], n_scaf)
    store = separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    got = _same_as_node_by_node(store)
    assert sum(r["split"] for r in got) >= G - 1


def test_tree_on_degenerate_columns_and_without_scgs(ctx):
    f = synth.make_features(9, 2500, 5, 2, n_lo=2, n_hi=40)
    keys = f["milli"].copy()
    N = keys.shape[1]
    rng = np.random.default_rng(3)
    keys[170] = 5000                                                   # constant, wide
    keys[171] = np.where(rng.random(N) < 0.5, 2000, 70000)             # two values, wide
    keys[172] = 1024 + rng.permutation(N)                              # all distinct, wide
    keys[173] = 7                                                      # constant, narrow
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], 2500)
    _same_as_node_by_node(separator.ColumnStore(ctx, keys, f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"]))
    _same_as_node_by_node(separator.ColumnStore(ctx, keys, f["dp2scaf"], f["scaf_len"]))           # no SCGs: the size gate
    # only narrow columns / only wide columns
    _same_as_node_by_node(separator.ColumnStore(ctx, keys[:160], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"]))
    _same_as_node_by_node(separator.ColumnStore(ctx, keys[179:], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"]))


def test_tree_with_more_scgs_than_the_lds_stage_holds(ctx):
    """above 256 SCGs the gate reads the lo / hi lists in place instead of a compacted copy in LDS"""
    f = synth.make_features(11, 4000, 6, 3, n_scg=300)
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], 4000)
    assert f["n_scg"] == 300
    got = _same_as_node_by_node(separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"]))
    assert sum(r["split"] for r in got) >= 3
