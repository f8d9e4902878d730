"""GPU parity of the separator (abw_sep_load / abw_sep_node) against golden vectors produced by the
reference's ClusterSeparatorBySensitivitySpecificity and against the CPU oracle."""
import glob
import os

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, separator, synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sep_*.npz")))


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _store(ctx, g):
    S = len(g["scaf_len"])
    off, ids = separator.scg_csr(g["scg_scaf"], g["scg_id"], S)
    return separator.ColumnStore(ctx, g["milli"], g["dp2scaf"], g["scaf_len"], off, ids, int(g["n_scg"])), off, ids


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
def test_binning_matches_reference(ctx, path):
    g = np.load(path)
    store, _o, _i = _store(ctx, g)
    nodes, s2c, d2c = separator.binning(store)
    ref = g["nodes"]
    assert len(nodes) == len(ref)
    for mine, r in zip(nodes, ref):
        assert (mine["id"], mine["ndps"], mine["split"], mine["dim"]) == (int(r[0]), int(r[1]), int(r[2]), int(r[3]))
        if mine["dim"] > 0:
            assert mine["key"] / 1000.0 == r[4]
        assert mine["spec"] == r[5] and mine["sens"] == r[6]                 # bit-exact doubles
        assert (mine["c1"], mine["n1"], mine["c2"], mine["n2"]) == tuple(int(v) for v in r[7:11])
    assert np.array_equal(s2c, g["scaf2cluster"])
    assert np.array_equal(d2c, g["dp2cluster"])


def _compare_node(store, scaf_ids, g_arrays, min_ndps=100):
    keys, dp2scaf, scaf_ndps, scaf_len, off, ids, n_scg = g_arrays
    sep = separator.ClusterSeparatorHip(store, scaf_ids)
    sep.cluster_ndps_threshold = min_ndps
    sep.separate()
    all_ids = np.arange(store.S, dtype=np.int32) if scaf_ids is None else scaf_ids
    dps = np.concatenate([np.arange(store.scaf_first[s], store.scaf_first[s] + store.scaf_ndps[s]) for s in all_ids]).astype(np.int32)
    res, per, raw, side = orc.separate(keys, dp2scaf, scaf_ndps, scaf_len, off, ids, n_scg, dps, min_ndps)
    mine = sep.per_dimension()
    assert np.array_equal(mine["found"], per["found"])
    f = per["found"] == 1
    assert np.array_equal(mine["key"][f], per["key"][f])
    assert np.array_equal(mine["sens"][f], per["sens"][f]) and np.array_equal(mine["spec"][f], per["spec"][f])
    assert (sep._res.split, sep._res.dimension, sep._res.key, sep._res.legal) == (res.split, res.dimension, res.key, res.legal)
    assert sep._res.sensitivity == res.sensitivity and sep._res.specificity == res.specificity
    assert (sep._res.ndps1, sep._res.ndps2) == (res.ndps1, res.ndps2)
    if res.legal and res.dimension > 0 or res.split:
        assert np.array_equal(sep._raw, raw)
    if res.split:
        assert np.array_equal(sep._side, side[all_ids])
    # the same node asked for its winner only (the reference's separate(); the library may then skip thresholds below the bar
    # another dimension set): same verdict, same bits
    only = separator.ClusterSeparatorHip(store, scaf_ids, per_dim=False)
    only.cluster_ndps_threshold = min_ndps
    only.separate()
    a, b = only._res, sep._res
    assert (a.split, a.dimension, a.key, a.legal, a.sensitivity, a.specificity, a.ndps1, a.ndps2) == \
           (b.split, b.dimension, b.key, b.legal, b.sensitivity, b.specificity, b.ndps1, b.ndps2)
    assert np.array_equal(only._side, sep._side) and np.array_equal(only._raw, sep._raw)
    return res


@pytest.mark.parametrize("path", GOLD[:3], ids=lambda p: os.path.basename(p)[:-4])
def test_per_dimension_winners_match_oracle(ctx, path):
    g = np.load(path)
    store, off, ids = _store(ctx, g)
    S = len(g["scaf_len"])
    scaf_ndps = np.bincount(g["dp2scaf"], minlength=S).astype(np.int32)
    arrays = (g["milli"], g["dp2scaf"], scaf_ndps, g["scaf_len"], off, ids, int(g["n_scg"]))
    _compare_node(store, None, arrays)
    rng = np.random.default_rng(1)
    for frac in (0.7, 0.4):
        sub = np.sort(rng.choice(S, size=int(S * frac), replace=False)).astype(np.int32)
        _compare_node(store, sub, arrays)
    # unordered scaffold list, lower size threshold
    sub = rng.permutation(S)[: S // 2].astype(np.int32)
    _compare_node(store, sub, arrays, min_ndps=20)


@pytest.mark.parametrize("seed,n_scaf,G,S_cov", [(1, 3000, 6, 3), (2, 12000, 12, 10)])
def test_synthetic_features_match_oracle(ctx, seed, n_scaf, G, S_cov):
    """make_features tables (incl. wide coverage columns with >1024 distinct values, big scaffolds)."""
    f = synth.make_features(seed, n_scaf, G, S_cov, n_lo=2, n_hi=14)
    S = n_scaf
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], S)
    store = separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    arrays = (f["milli"], f["dp2scaf"], f["scaf_ndps"], f["scaf_len"], off, ids, f["n_scg"])
    res = _compare_node(store, None, arrays)
    assert res.split == 1
    # and the whole recursion gives the same bins as the oracle's
    nodes, s2c, d2c = separator.binning(store, max_nodes=9)
    onodes, os2c, od2c = orc.recursive_bins(f["milli"], f["dp2scaf"], f["scaf_ndps"], f["scaf_len"], off, ids, f["n_scg"], max_nodes=9)
    assert [(a["id"], a["split"], a["dim"], a["key"], a["n1"], a["n2"]) for a in nodes] == \
           [(b["id"], b["split"], b["dim"], b["key"], b["n1"], b["n2"]) for b in onodes]
    assert np.array_equal(s2c, os2c) and np.array_equal(d2c, od2c)


def test_scaffolds_of_thousands_of_dps_match_oracle(ctx):
    """Scaffolds of up to 4 000 dps (a tile of the narrow accumulation then holds more than 65 535 dps; the order statistic
    of a scaffold takes its long form), next to a table of ordinary ones."""
    for seed, n_scaf, n_hi in ((31, 40, 4000), (32, 400, 60)):
        f = synth.make_features(seed, n_scaf, 2, 2, n_lo=2, n_hi=n_hi)
        assert (f["scaf_ndps"].sum() > 65535) == (n_hi == 4000)
        off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], n_scaf)
        store = separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
        arrays = (f["milli"], f["dp2scaf"], f["scaf_ndps"], f["scaf_len"], off, ids, f["n_scg"])
        _compare_node(store, None, arrays)
        sub = np.sort(np.random.default_rng(seed).choice(n_scaf, size=n_scaf * 3 // 4, replace=False)).astype(np.int32)
        _compare_node(store, sub, arrays)                            # the scattered form of the same kernel
        want, s2c, _d2c = separator.binning(store)
        got, s2c_t, _dt = separator.binning_tree(store)
        key = lambda r: (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])
        assert [key(r) for r in got] == [key(r) for r in want] and np.array_equal(s2c_t, s2c)


def test_load_rejects_bad_layout(ctx):
    keys = np.zeros((2, 6), dtype=np.int32)
    with pytest.raises(_lib.AbwError):      # singleton scaffold
        separator.ColumnStore(ctx, keys, np.array([0, 0, 1, 2, 2, 2]), np.ones(3, np.int64))
    with pytest.raises(_lib.AbwError):      # not contiguous
        separator.ColumnStore(ctx, keys, np.array([0, 1, 0, 1, 2, 2]), np.ones(3, np.int64))


def test_degenerate_columns_match_oracle(ctx):
    """Columns that stress the run handling of the sorted walk and the LDS histograms: a constant wide column (one bin:
    every tile's run is open on both sides), a two-valued wide column, a wide column of all-distinct keys, a constant
    narrow column; scaffolds of 2..40 dps so that both order-statistic paths run."""
    f = synth.make_features(9, 2500, 5, 2, n_lo=2, n_hi=40)
    keys = f["milli"].copy()
    N = keys.shape[1]
    rng = np.random.default_rng(3)
    keys[170] = 5000                                                   # constant, wide
    keys[171] = np.where(rng.random(N) < 0.5, 2000, 70000)             # two values, wide
    keys[172] = 1024 + rng.permutation(N)                              # all distinct, wide
    keys[173] = 7                                                      # constant, narrow
    S = 2500
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], S)
    store = separator.ColumnStore(ctx, keys, f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    arrays = (keys, f["dp2scaf"], f["scaf_ndps"], f["scaf_len"], off, ids, f["n_scg"])
    _compare_node(store, None, arrays)
    sub = np.sort(rng.choice(S, size=S // 10, replace=False)).astype(np.int32)      # small node: the scatter path
    _compare_node(store, sub, arrays, min_ndps=20)


def test_configs0_whole_recursion_matches_the_reference(ctx, golden_dir):
    """BASELINE configs[0]: the 10 000-scaffold table, the whole recursion.  tests/golden/recursion_c1.npz holds what the
    reference's own separator (oracle/_ref/ref_sep_driver: reference objects, reference main loop) printed for it --
    49 nodes with "%.17g" doubles, and scaf2cluster; the table is regenerated from its seed."""
    g = np.load(os.path.join(golden_dir, "recursion_c1.npz"))
    f = synth.make_features(20240, 10000, 25, 3)
    assert tuple(g["shape"]) == f["milli"].shape and int(g["checksum"][0]) == int(f["milli"].astype(np.int64).sum())
    S = len(f["scaf_len"])
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], S)
    store = separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    nodes, s2c, _d2c = separator.binning(store)
    ref = g["nodes"]
    assert len(nodes) == len(ref) == 49
    for a, r in zip(nodes, ref):
        assert (a["id"], a["ndps"], a["split"], a["dim"]) == (int(r[0]), int(r[1]), int(r[2]), int(r[3]))
        assert a["dim"] <= 0 or a["key"] / 1000.0 == r[4]
        assert a["spec"] == r[5] and a["sens"] == r[6]                          # the reference's doubles, bit for bit
        assert (a["c1"], a["n1"], a["c2"], a["n2"]) == (int(r[7]), int(r[8]), int(r[9]), int(r[10]))
    assert np.array_equal(s2c, g["scaf2cluster"])


@pytest.mark.parametrize("path", GOLD[:2], ids=lambda p: os.path.basename(p)[:-4])
def test_unpacked_accumulation_gives_the_same_recursion(path, monkeypatch):
    """The narrow accumulation has two forms: counters packed two to an LDS word (tables whose scaffolds have at most 63 dps) and
    one counter per word.  The goldens run packed everywhere else; here the other form, against the reference's nodes."""
    monkeypatch.setenv("ABW_SEP_UNPACKED", "1")
    c = _lib.Context(0)
    try:
        g = np.load(path)
        store, _o, _i = _store(c, g)
        nodes, s2c, _d2c = separator.binning(store)
        assert [(n["id"], n["ndps"], n["split"], n["dim"], n["spec"], n["sens"]) for n in nodes] == \
               [(int(r[0]), int(r[1]), int(r[2]), int(r[3]), r[5], r[6]) for r in g["nodes"]]
        assert np.array_equal(s2c, g["scaf2cluster"])
        rows, s2c_t, _dt = separator.binning_tree(store)
        assert [(r["id"], r["split"], r["dim"], r["key"]) for r in rows] == [(n["id"], n["split"], n["dim"], n["key"]) for n in nodes]
        assert np.array_equal(s2c_t, s2c)
    finally:
        c.close()
