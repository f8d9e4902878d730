"""GPU parity of the scaffold-sharded separator (abw_sep_shard_*): 2-4 shards of a table, played from one
process on one GPU, give exactly what abw_sep_node gives on the whole table (which the other tests pin to the
reference's golden vectors and to the oracle) -- per-dimension winners, global winner, sides, whole recursion."""
import glob
import os

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, multi, separator, synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sep_*.npz")))


def _shards(keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, groups):
    parts = [separator.carve_shard(keys, dp2scaf, scaf_len, g, scg_scaf, scg_id) for g in groups]
    dv, do = multi.merge_dictionaries([p["keys"] for p in parts])
    ctxs, stores = [], []
    for p in parts:
        c = _lib.Context(0)
        off, ids = separator.scg_csr(p["scg_scaf"], p["scg_id"], len(p["scaf_len"]))
        stores.append(separator.ShardedColumnStore(c, p["keys"], p["dp2scaf"], p["scaf_len"], dv, do, off, ids, n_scg))
        ctxs.append(c)
    return ctxs, stores, parts


def _check_against_whole(keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, groups, max_nodes=1 << 30):
    S = len(scaf_len)
    ctx = _lib.Context(0)
    off, ids = separator.scg_csr(scg_scaf, scg_id, S)
    whole = separator.ColumnStore(ctx, keys, dp2scaf, scaf_len, off, ids, n_scg)
    ctxs, stores, parts = _shards(keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, groups)
    try:
        # root node: per-dimension winners, raw sides and scaffold sides
        sep = separator.ClusterSeparatorHip(whole)
        sep.separate()
        nodes = separator.separate_shards_local(stores, [None] * len(stores))
        for nd in nodes:
            assert np.array_equal(nd.per, sep.per_dimension())
            o, r = nd.out, sep._res
            assert (o.split, o.dimension, o.key, o.legal, o.ndps1, o.ndps2) == (r.split, r.dimension, r.key, r.legal, r.ndps1, r.ndps2)
            assert o.sensitivity == r.sensitivity and o.specificity == r.specificity
        if sep._res.legal and sep._res.dimension > 0:
            for nd, g, p in zip(nodes, groups, parts):
                assert np.array_equal(nd.raw, sep._raw[p["dps"]])
        if sep._res.split:
            for nd, g in zip(nodes, groups):
                assert np.array_equal(nd.side, sep._side[np.asarray(g)])
        # whole recursion
        want, s2c, _d2c = separator.binning(whole, max_nodes=max_nodes)
        rows, s2c_parts = separator.binning_shards_local(stores, max_nodes=max_nodes)
        key = lambda a: (a["id"], a["ndps"], a["split"], a["dim"], a["key"], a["spec"], a["sens"], a["c1"], a["n1"], a["c2"], a["n2"])
        assert [key(a) for a in rows] == [key(b) for b in want]
        got = np.zeros(S, dtype=np.int32)
        for g, a in zip(groups, s2c_parts):
            got[np.asarray(g)] = a
        assert np.array_equal(got, s2c)
        # the exchange above was the library's (abw_sep_shard_combine); plain tensor arithmetic in its place gives the same
        rows_t, s2c_t = separator.binning_shards_local(stores, max_nodes=max_nodes, combine="torch")
        assert [key(a) for a in rows_t] == [key(b) for b in want] and all(np.array_equal(a, b) for a, b in zip(s2c_t, s2c_parts))
        return rows, sep
    finally:
        for c in ctxs + [ctx]:
            c.close()


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("world", [2, 3])
def test_golden_sets_sharded(path, world):
    g = np.load(path)
    S = len(g["scaf_len"])
    b = multi.scaffold_bounds(np.bincount(g["dp2scaf"], minlength=S), world)
    groups = [np.arange(b[k], b[k + 1]) for k in range(world)]
    rows, _ = _check_against_whole(g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]), groups)
    ref = g["nodes"]                                     # and, transitively, the reference's own nodes
    assert [(r["id"], r["split"], r["dim"]) for r in rows] == [(int(r[0]), int(r[2]), int(r[3])) for r in ref]


def test_wide_columns_interleaved_shards_and_empty_shards():
    """Coverage columns with > 1024 distinct keys (dictionary bins), scaffolds dealt round-robin to 4 shards,
    then a layout in which whole subtrees live on one shard (the others pass empty nodes)."""
    f = synth.make_features(3, 6000, 8, 5, n_lo=2, n_hi=12)
    S = 6000
    args = (f["milli"], f["dp2scaf"], f["scaf_len"], f["scg_scaf"], f["scg_id"], f["n_scg"])
    rows, sep = _check_against_whole(*args, [np.arange(k, S, 4) for k in range(4)], max_nodes=9)
    assert rows[0]["split"] == 1
    side = sep._side                                     # shard 0 = cluster1 of the root, shard 1 = cluster2
    rows, _ = _check_against_whole(*args, [np.flatnonzero(side == 1), np.flatnonzero(side == 2)], max_nodes=9)
    assert any(r["empty_shards"] > 0 for r in rows[1:])


def test_shard_load_rejects_bad_dictionaries():
    c = _lib.Context(0)
    try:
        keys = np.array([[1, 2, 3, 4], [5000, 6000, 5000, 7000]], dtype=np.int32)
        dp2scaf, slen = np.array([0, 0, 1, 1]), np.ones(2, np.int64)
        ok_v, ok_o = np.array([5000, 6000, 7000, 9000], np.int32), np.array([0, 0, 4], np.int64)
        separator.ShardedColumnStore(c, keys, dp2scaf, slen, ok_v, ok_o)
        with pytest.raises(_lib.AbwError):               # wide keys without a dictionary
            separator.ShardedColumnStore(c, keys, dp2scaf, slen, np.zeros(0, np.int32), np.array([0, 0, 0], np.int64))
        with pytest.raises(_lib.AbwError):               # a key missing from the dictionary
            separator.ShardedColumnStore(c, keys, dp2scaf, slen, np.array([5000, 6000], np.int32), np.array([0, 0, 2], np.int64))
        with pytest.raises(_lib.AbwError):               # not ascending
            separator.ShardedColumnStore(c, keys, dp2scaf, slen, np.array([6000, 5000, 7000], np.int32), np.array([0, 0, 3], np.int64))
        with pytest.raises(_lib.AbwError):               # eval without accum
            st = separator.ShardedColumnStore(c, keys, dp2scaf, slen, ok_v, ok_o)
            out = _lib.Split()
            c.check(_lib.lib().abw_sep_shard_eval(c.handle, 100, out, None, None, None))
        with pytest.raises(_lib.AbwError):               # combine without accum
            import ctypes as C
            c.check(_lib.lib().abw_sep_shard_combine((C.c_void_p * 1)(c.handle), 1))
    finally:
        c.close()
