"""The recursion with the table's DIMENSIONS spread over several contexts (abw_sep_tree_begin / eval / assign / advance,
SURVEY section 8(e)): every context holds all scaffolds and some of the columns, per level the contexts exchange their best
threshold of every node and the sides the winners' owners decided.  Same nodes and bins as one context with the whole
table -- which the other tests pin to the reference's goldens -- for 2, 3 and 5 contexts played from one process."""
import glob
import os

import numpy as np
import pytest

from abawaca_amd import _lib, separator, synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sep_*.npz")))


def _key(r):
    return (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])


def _check(keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, world, max_nodes=0, parts=None):
    S = len(scaf_len)
    off, ids = separator.scg_csr(scg_scaf, scg_id, S)
    ctx = _lib.Context(0)
    ctxs = [_lib.Context(0) for _ in range(world)]
    try:
        want, s2c, _dt = separator.binning_tree(separator.ColumnStore(ctx, keys, dp2scaf, scaf_len, off, ids, n_scg), max_nodes=max_nodes)
        parts = parts if parts is not None else separator.dim_partition(keys, world)
        assert sorted(np.concatenate(parts).tolist()) == list(range(keys.shape[0]))
        stores = [separator.DimShardStore(c, keys, p, dp2scaf, scaf_len, off, ids, n_scg) for c, p in zip(ctxs, parts)]
        res, exchanged = separator.binning_tree_dims_local(stores, max_nodes=max_nodes)
        for rows, s2c_k in res:                                  # every context walked the same tree
            assert [_key(r) for r in rows] == [_key(r) for r in want]
            assert np.array_equal(s2c_k, s2c)
        return want, exchanged
    finally:
        for c in ctxs + [ctx]:
            c.close()


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("world", [2, 3])
def test_golden_sets_dimension_sharded(path, world):
    g = np.load(path)
    want, _ = _check(g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]), world)
    ref = g["nodes"]                                             # and, transitively, the reference's own nodes
    assert [(r["id"], r["split"], r["dim"]) for r in want] == [(int(r[0]), int(r[2]), int(r[3])) for r in ref]


def test_wide_columns_five_contexts_and_a_cut_work_list():
    f = synth.make_features(3, 6000, 8, 5, n_lo=2, n_hi=12)
    args = (f["milli"], f["dp2scaf"], f["scaf_len"], f["scg_scaf"], f["scg_id"], f["n_scg"])
    want, exchanged = _check(*args, 5)
    assert sum(r["split"] for r in want) >= 7
    assert exchanged < 40 * (6000 + 64 * 40)                      # a byte per scaffold and a record per node, per level
    _check(*args, 3, max_nodes=5)
    # ties between contexts: the same column on two contexts (equal products) -- the smaller dimension wins, as in one table
    keys2 = np.concatenate([f["milli"], f["milli"][180:182]])
    parts = [np.arange(0, keys2.shape[0] - 2), np.arange(keys2.shape[0] - 2, keys2.shape[0])]
    _check(keys2, *args[1:], 2, parts=parts)


def test_calls_out_of_order_are_refused():
    f = synth.make_features(4, 800, 3, 2)
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], 800)
    c = _lib.Context(0)
    try:
        st = separator.DimShardStore(c, f["milli"], np.arange(f["milli"].shape[0]), f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
        with pytest.raises(_lib.AbwError):
            c.check(_lib.lib().abw_sep_tree_advance(c.handle))   # nothing begun
        run = separator.DimShardRun(st)
        with pytest.raises(_lib.AbwError):
            run.advance()                                        # not evaluated
        best = run.eval()
        with pytest.raises(_lib.AbwError):
            run.eval()                                           # the level waits for assign
        wrong = best.copy()
        wrong["key"] += 1
        with pytest.raises(_lib.AbwError):
            run.assign(wrong, np.ones(len(best), bool))          # not the record eval returned
        run.assign(best, np.ones(len(best), bool))
        run.advance()
        # the level lives in the context's work buffers between the calls: another entry point on the SAME context in between
        # ends the recursion with an error instead of letting it go on with buffers that may have been freed or overwritten
        best = run.eval()
        from abawaca_amd import coverage
        coverage.accumulate(c, [0], [1], [100], 1, [0], [5], [50])
        with pytest.raises(_lib.AbwError, match="another entry point"):
            run.assign(best, np.ones(len(best), bool))
        with pytest.raises(_lib.AbwError):
            run.advance()
        run = separator.DimShardRun(st)                          # begun again: fine
        best = run.eval()
        run.assign(best, np.ones(len(best), bool))
        run.advance()
        # the most likely interleaved call -- looking at a node of the same table with abw_sep_node, which reserves none of
        # the device work buffers but writes the pinned block the level's host records live in -- is noticed, too
        best = run.eval()
        sep = separator.ClusterSeparatorHip(st)
        sep.separate()
        with pytest.raises(_lib.AbwError, match="another entry point"):
            run.assign(best, np.ones(len(best), bool))
        run = separator.DimShardRun(st)
        best = run.eval()
        run.assign(best, np.ones(len(best), bool))
        sep.separate()
        with pytest.raises(_lib.AbwError, match="another entry point"):
            run.advance()
        rows, _s2c, _dt = separator.binning_tree(separator.ColumnStore(c, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"]))
        assert len(rows) >= 1
    finally:
        c.close()
