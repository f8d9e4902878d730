"""The recursion with the SCAFFOLDS of the signature columns sharded over several ranks and the coverage columns spread by
dimension (abw_sep_stree_* + the level calls; SURVEY section 8(e) "separator sweep (B)", BASELINE configs[4]): same nodes and
bins as one context with the whole table -- which the other tests pin to the reference's goldens -- for 2, 3 and 8 ranks played
from one process, and the bytes a rank exchanges per level stay small."""
import glob
import os

import numpy as np
import pytest

from abawaca_amd import _lib, separator, synth

pytestmark = pytest.mark.gpu
GOLD = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "sep_*.npz")))


def _key(r):
    return (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])


def _check(keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, world, max_nodes=0):
    S = len(scaf_len)
    off, ids = separator.scg_csr(scg_scaf, scg_id, S)
    ctx = _lib.Context(0)
    pairs = [(_lib.Context(0), _lib.Context(0)) for _ in range(world)]
    try:
        want, s2c, _dt = separator.binning_tree(separator.ColumnStore(ctx, keys, dp2scaf, scaf_len, off, ids, n_scg), max_nodes=max_nodes)
        shards = separator.make_scaf_shards(pairs, keys, dp2scaf, scaf_len, off, ids, n_scg)
        assert sum(sh.s1 - sh.s0 for sh in shards) == S
        res, exchanged, levels = separator.binning_stree_local(shards, max_nodes=max_nodes)
        for rows, s2c_k in res:                                  # every rank walked the same tree
            assert [_key(r) for r in rows] == [_key(r) for r in want]
            assert np.array_equal(s2c_k, s2c)
        return want, exchanged, levels
    finally:
        for a, b in pairs:
            a.close(); b.close()
        ctx.close()


@pytest.mark.parametrize("path", GOLD, ids=lambda p: os.path.basename(p)[:-4])
@pytest.mark.parametrize("world", [2, 3])
def test_golden_sets_scaffold_sharded(path, world):
    g = np.load(path)
    want, _ex, _lv = _check(g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]), world)
    ref = g["nodes"]                                             # and, transitively, the reference's own nodes
    assert [(r["id"], r["split"], r["dim"]) for r in want] == [(int(r[0]), int(r[2]), int(r[3])) for r in ref]


def test_eight_ranks_wide_columns_and_a_cut_work_list():
    f = synth.make_features(3, 6000, 8, 5, n_lo=2, n_hi=12)
    args = (f["milli"], f["dp2scaf"], f["scaf_len"], f["scg_scaf"], f["scg_id"], f["n_scg"])
    want, exchanged, levels = _check(*args, 8)                    # more ranks than coverage columns: three walk a constant column
    assert sum(r["split"] for r in want) >= 7
    assert exchanged <= levels * (8 << 20)
    _check(*args, 3, max_nodes=5)


def test_calls_out_of_order_and_wrong_tables_are_refused():
    f = synth.make_features(4, 800, 3, 2)
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], 800)
    c = _lib.Context(0)
    try:
        L = _lib.lib()
        with pytest.raises(_lib.AbwError):
            c.check(L.abw_sep_stree_begin(c.handle, 0))          # nothing loaded
        separator.ColumnStore(c, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
        with pytest.raises(_lib.AbwError, match="narrow columns only"):
            c.check(L.abw_sep_stree_begin(c.handle, 0))          # the table holds coverage columns
        separator.ColumnStore(c, f["milli"][:179], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
        c.check(L.abw_sep_stree_begin(c.handle, 0))
        best = np.zeros(4, dtype=separator._WIN_DTYPE)
        import ctypes as C
        with pytest.raises(_lib.AbwError):
            c.check(L.abw_sep_stree_eval(c.handle, 100, C.c_void_p(best.ctypes.data), 4))      # nothing accumulated
    finally:
        c.close()
