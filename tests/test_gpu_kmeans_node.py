"""abw_km_load / abw_km_node (k-means separator on the resident table) against the same node done the long
way: the node's columns gathered on the host, abw_km_run, and the finalisation of reference
src/ClusterSeparator.cpp:49-53,94-132 in numpy -- which is what integration/ClusterSeparatorKMeans.cpp did
before the table became resident.  The restarts themselves are pinned to the CPU oracle elsewhere
(tests/test_gpu_kmeans.py); the k-means has no counterpart in the reference (parity unpinned)."""
import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, kmeans, synth

pytestmark = pytest.mark.gpu


def _long_way(ctx, x, dp2scaf, ids, R, seed, iters, min_ndps, chunk=0):
    S = int(dp2scaf.max()) + 1
    ndps = np.bincount(dp2scaf, minlength=S)
    first = np.concatenate([[0], np.cumsum(ndps)[:-1]])
    ids = np.arange(S) if ids is None else ids
    dps = np.concatenate([np.arange(first[s], first[s] + ndps[s]) for s in ids])
    n = len(dps)
    if n < 2 * min_ndps or n < 32:
        return dict(best=-1, split=0, n=n)
    xa = np.ascontiguousarray(x[:, dps])
    assign, _c, inertia = kmeans.kmeans_restarts(ctx, xa, R, seed, iters, chunk)
    oa, _oc, oi = orc.kmeans(xa, R, seed, iters, chunk or kmeans.auto_chunk(ctx, n, R))
    assert np.array_equal(assign, oa) and np.array_equal(inertia, oi)          # the restarts are the oracle's
    best = int(np.argmin(inertia))
    side = assign[best]
    ones = int(side.sum())
    firstside = 1 if ones < n - ones else 0
    in1 = side == firstside
    raw1 = int(in1.sum())
    if raw1 < min_ndps:
        return dict(best=best, split=0, n=n, raw1=raw1, inertia=inertia[best], nosplit=True)
    sc = dp2scaf[dps]
    k = np.bincount(sc, weights=in1, minlength=S)[ids].astype(np.int64)
    tot = ndps[ids]
    to1 = 2 * k >= tot
    agree = int(np.where(to1, k, tot - k).sum())
    n1, n2 = int(tot[to1].sum()), int(tot[~to1].sum())
    return dict(best=best, split=int(n1 >= min_ndps and n2 >= min_ndps), n=n, raw1=raw1, inertia=inertia[best],
                agreement=agree / n, n1=n1, n2=n2, side=np.where(to1, 1, 2).astype(np.uint8), raw=np.where(in1, 1, 2).astype(np.uint8))


@pytest.mark.parametrize("seed,n_scaf,G", [(4, 1500, 2), (5, 2500, 3)])
def test_node_matches_the_long_way(seed, n_scaf, G):
    f = synth.make_features(seed, n_scaf, G, 3)
    keys = f["milli"].astype(np.float64)
    x = ((keys - keys.mean(axis=1, keepdims=True)) / np.maximum(keys.std(axis=1, keepdims=True), 1e-6)).astype(np.float32)
    ctx = _lib.Context(0)
    try:
        store = kmeans.KmColumnStore(ctx, x, f["dp2scaf"])
        rng = np.random.default_rng(seed)
        nodes = [None, np.sort(rng.choice(n_scaf, size=n_scaf // 2, replace=False)).astype(np.int32),
                 np.sort(rng.choice(n_scaf, size=n_scaf // 7, replace=False)).astype(np.int32), np.arange(10, dtype=np.int32)]
        for ids in nodes:
            for min_ndps in (100, 10 ** 6 if ids is None else 100):
                out, side, raw = kmeans.km_node(store, ids, 8, 77, 6, min_ndps)
                want = _long_way(ctx, x, f["dp2scaf"], ids, 8, 77, 6, min_ndps)
                assert out.ndps == want["n"] and out.best_restart == want["best"] and out.split == want["split"]
                assert out.chunk == (kmeans.auto_chunk(ctx, want["n"], 8) if want["best"] >= 0 else 0)      # what chunk = 0 became
                if want["best"] < 0 or want.get("nosplit"):
                    assert not side.any() and not raw.any()
                    continue
                assert out.inertia == want["inertia"] and out.raw1 == want["raw1"]
                assert (out.ndps1, out.ndps2) == (want["n1"], want["n2"]) and out.agreement == want["agreement"]
                assert np.array_equal(side, want["side"]) and np.array_equal(raw, want["raw"])
        out, side, raw = kmeans.km_node(store, None, 8, 77, 6, 100)
        assert out.split == 1 and out.agreement > 0.8                           # planted genomes come apart
        # an explicit chunk: the node is the oracle's run with THAT chunk (reproducible whatever the device would pick)
        for chunk in (96, 1024):
            o2, s2, r2 = kmeans.km_node(store, nodes[1], 8, 77, 6, 100, chunk=chunk)
            w2 = _long_way(ctx, x, f["dp2scaf"], nodes[1], 8, 77, 6, 100, chunk=chunk)
            assert o2.chunk == chunk and o2.inertia == w2["inertia"] and np.array_equal(r2, w2["raw"]) and np.array_equal(s2, w2["side"])
        with pytest.raises(_lib.AbwError):
            kmeans.km_node(store, None, 8, 77, 6, 100, chunk=100)                # not a multiple of 32
        with pytest.raises(_lib.AbwError):
            kmeans.km_node(store, np.array([5, 3], dtype=np.int32), 8, 77, 6)    # not ascending
    finally:
        ctx.close()


def test_auto_chunk_is_the_exported_rule_for_this_device():
    """abw_km_auto_chunk(ctx, n, R) == abw_km_chunk_for(the device's compute units, n, R): the adapter's device-independent
    default (256 units) is the library's own rule, not a copy of it."""
    import torch
    ctx = _lib.Context(0)
    try:
        cu = torch.cuda.get_device_properties(0).multi_processor_count
        L = _lib.lib()
        for n in (32, 4097, 200_000, 799_859, 3_999_008):
            for R in (1, 16, 32, 64, 100, 256):
                assert kmeans.auto_chunk(ctx, n, R) == L.abw_km_chunk_for(cu, n, R)
    finally:
        ctx.close()
