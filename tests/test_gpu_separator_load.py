"""abw_sep_load builds the column store with kernels (columns classified, the wide ones sorted and ranked on the device).
The host build it replaced is still in the library (it serves the shards, whose layout comes from shared dictionaries) and
is switched on for any table by ABW_SEP_LOAD_HOST=1: both builds, and abw_sep_load_dev from keys that are already on the
device, must give the same per-dimension winners at the root and the same whole recursion."""
import os

import numpy as np
import pytest

from abawaca_amd import _lib, separator, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _everything(store):
    sep = separator.ClusterSeparatorHip(store)
    sep.separate()
    per = sep.per_dimension().tobytes()                     # found, key, sensitivity and specificity bits of every dimension
    rows, s2c, _dt = separator.binning_tree(store)
    key = lambda r: (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])
    return per, [key(r) for r in rows], np.asarray(s2c).copy()


def _host_build(ctx, *args):
    os.environ["ABW_SEP_LOAD_HOST"] = "1"
    try:
        return separator.ColumnStore(ctx, *args)
    finally:
        del os.environ["ABW_SEP_LOAD_HOST"]


@pytest.mark.parametrize("seed,n_scaf,G,S_cov,n_hi", [(5, 4000, 8, 3, 12), (6, 20000, 15, 10, 6), (7, 900, 3, 20, 30)])
def test_device_build_equals_host_build(ctx, seed, n_scaf, G, S_cov, n_hi):
    import torch
    f = synth.make_features(seed, n_scaf, G, S_cov, n_lo=2, n_hi=n_hi)
    keys = np.ascontiguousarray(f["milli"], dtype=np.int32).copy()
    rng = np.random.default_rng(seed)
    # what the generator does not make: a wide column with negative keys and few distinct values, a narrow one of a single
    # value, a column that is wide by ONE key only
    keys[0] = rng.integers(-50, 50, size=keys.shape[1]) * 1000
    keys[1] = 7
    keys[2, rng.integers(0, keys.shape[1])] = 1024
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], n_scaf)
    args = (keys, f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    want = _everything(_host_build(ctx, *args))
    got = _everything(separator.ColumnStore(ctx, *args))
    assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])
    t = torch.from_numpy(keys).to("cuda:0")
    dev = _everything(separator.ColumnStore(ctx, None, *args[1:], device_keys=(t.data_ptr(), keys.shape[0], keys.shape[1])))
    assert dev[0] == want[0] and dev[1] == want[1] and np.array_equal(dev[2], want[2])
    assert any(r[2] for r in want[1])                        # the recursion does split


def test_a_table_without_wide_columns(ctx):
    f = synth.make_features(8, 2500, 5, 0, n_lo=2, n_hi=9)
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], 2500)
    args = (f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    assert int(np.max(f["milli"])) < 1024
    want, got = _everything(_host_build(ctx, *args)), _everything(separator.ColumnStore(ctx, *args))
    assert got[0] == want[0] and got[1] == want[1] and np.array_equal(got[2], want[2])
