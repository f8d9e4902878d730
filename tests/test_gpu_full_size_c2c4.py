"""BASELINE.json's two big configurations at FULL size on one GPU, with the node's partitions played by several contexts
on the one device (the driver's 8-GPU run cannot be rehearsed here; the partitions are exact, so shards on one device
prove the same thing).  The CPU oracle would take hours at these sizes: size-independent properties instead.

  configs[2]  1M scaffolds -> ~4.0 M dps x 199 dims, ONE job of 256 restarts
      * "restarts" partition: 8 ranks x 32 restarts with shifted seeds == the 256 at once, bit for bit
      * "points" partition: 2, 4 and 8 point shards -- every shard ends with the same centroid / inertia bits; against
        the unsharded run the fp32 partial sums are cut at other places, so the results agree to a stated tolerance and
        (almost) every point keeps its cluster
  configs[4]  2M scaffolds -> ~8.0 M dps x 199 dims, 30 SCGs x 400 genomes, the whole recursion
      * node by node, in one call (abw_sep_tree), and with the dimensions spread over 2 and 8 contexts: the same nodes and bins
      * unsharded: the tree is well-formed, every scaffold ends in exactly one bin, the root's winner recomputed from the
        definition in numpy gives the returned doubles, the planted genomes come out
      * 2 and 8 scaffold shards: the same nodes (winner dimension, key, sensitivity / specificity bits, children sizes)
        and the same bins as unsharded
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from abawaca_amd import _lib, kmeans, multi, separator  # noqa: E402

pytestmark = pytest.mark.gpu


def test_configs2_256_restarts_under_both_partitions():
    import torch
    dev = torch.device("cuda", 0)
    c2 = bench.CONFIGS[2]
    keys_t, _dp2scaf, _len, _genome, _G = bench.make_points(20242, c2["scaffolds"], c2["samples"], dev, c2["genomes"])
    x_t = bench.standardise(keys_t)
    torch.cuda.synchronize()          # the library runs on its own stream: x_t must be complete before it reads it
    del keys_t
    D, N = x_t.shape
    assert D == 199 and N > 3_900_000
    R, iters, seed = c2["restarts"], 3, 20240
    ctx = _lib.Context(0)
    ctxs = []
    try:
        chunk = 1024                                              # one chunk for every run of this test: chunk is part of the bits
        a = torch.empty((R, N), dtype=torch.uint8, device=dev)
        c = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
        i = torch.empty(R, dtype=torch.float64, device=dev)
        kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R, seed, iters, chunk, a, c, i)
        ctx.synchronize()
        ones = a.sum(dim=1, dtype=torch.int64)
        assert int(ones.min()) > 0 and int(ones.max()) < N
        # ---- restarts partition: rank k of 8 runs restarts [32k, 32k + 32) with base_seed + 32k
        world = 8
        a8 = torch.empty((R // world, N), dtype=torch.uint8, device=dev)
        c8 = torch.empty((R // world, 2, D), dtype=torch.float32, device=dev)
        i8 = torch.empty(R // world, dtype=torch.float64, device=dev)
        for k in range(world):
            first, n = multi.restart_range(R, k, world)
            kmeans.kmeans_restarts_dev(ctx, x_t, N, D, n, multi.rank_seed(seed, first), iters, chunk, a8, c8, i8)
            ctx.synchronize()
            assert torch.equal(i8, i[first:first + n]) and torch.equal(c8.view(torch.int32), c[first:first + n].view(torch.int32))
            assert torch.equal(a8, a[first:first + n])
        del a8
        winner = multi.best_restart(i.cpu(), 0)
        # ---- points partition: 2, 4, 8 shards as contexts on this device
        for world in (2, 4, 8):
            b = multi.shard_bounds(N, world)
            while len(ctxs) < world:
                ctxs.append(_lib.Context(0))
            xs = [x_t[:, b[k]:b[k + 1]].contiguous() for k in range(world)]
            torch.cuda.synchronize()                              # the contexts run on their own streams
            runs = [kmeans.ShardRun(ctxs[k], xs[k], b[k + 1] - b[k], b[k], N, D, R, seed, chunk) for k in range(world)]

            def sync():
                for cc in ctxs[:world]:
                    cc.synchronize()
                torch.cuda.synchronize()

            def combine(name):
                sync()
                bufs = [getattr(r, name) for r in runs]
                tot = bufs[0].clone()
                for t in bufs[1:]:
                    tot += t                                      # shard order, as multi.combine_ordered_sum does over RCCL
                for t in bufs:
                    t.copy_(tot)
                sync()
            combine("init")
            combine("moments")
            for r in runs:
                r.ready()
            asg = [torch.empty((R, b[k + 1] - b[k]), dtype=torch.uint8, device=dev) for k in range(world)]
            for it in range(iters + 1):
                for k, r in enumerate(runs):
                    r.step(asg[k] if it == iters else None)
                combine("sums")
                for r in runs:
                    r.update(it == iters)
            cents = [torch.empty((R, 2, D), dtype=torch.float32, device=dev) for _ in range(world)]
            inert = [torch.empty(R, dtype=torch.float64, device=dev) for _ in range(world)]
            for k, r in enumerate(runs):
                r.finish(cents[k], inert[k])
            sync()
            for k in range(1, world):                             # identical bits on every shard
                assert torch.equal(inert[k], inert[0]) and torch.equal(cents[k].view(torch.int32), cents[0].view(torch.int32))
            # against the unsharded run: same algorithm, fp32 partial sums cut at the shard boundaries
            assert float(((inert[0] - i).abs() / i).max()) < 1e-6
            assert float((cents[0] - c).abs().max()) < 1e-3
            flipped = sum(int((asg[k] != a[:, b[k]:b[k + 1]]).sum()) for k in range(world))
            assert flipped <= R * N * 1e-6                        # near-tie points only
            assert multi.best_restart(inert[0].cpu(), 0)[0] == winner[0]
            del runs, xs, asg
    finally:
        ctx.close()
        for cc in ctxs:
            cc.close()


def _tree_key(r):
    return (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])


def test_configs4_whole_recursion_unsharded_and_scaffold_sharded():
    import torch
    dev = torch.device("cuda", 0)
    c4 = bench.CONFIGS[4]
    keys_t, dp2scaf, scaf_len, genome, G = bench.make_points(20244, c4["scaffolds"], c4["samples"], dev, c4["genomes"])
    keys = keys_t.cpu().numpy()
    del keys_t
    torch.cuda.empty_cache()
    D, N = keys.shape
    S = len(scaf_len)
    assert S == 2_000_000 and D == 199
    scg_scaf, scg_id = bench.make_scgs(20244, genome, G)
    so, si = separator.scg_csr(scg_scaf, scg_id, S)
    ndps = np.bincount(dp2scaf, minlength=S).astype(np.int64)
    ctx = _lib.Context(0)
    try:
        store = separator.ColumnStore(ctx, keys, dp2scaf, scaf_len, so, si, 30)
        nodes, s2c, d2c = separator.binning(store)
        # the same recursion in one call (abw_sep_tree), and with the DIMENSIONS spread over 2 and 8 contexts
        rows_t, s2c_t, _dt = separator.binning_tree(store)
        assert [_tree_key(r) for r in rows_t] == [_tree_key(r) for r in nodes] and np.array_equal(s2c_t, s2c)
    finally:
        ctx.close()
    for world in (2, 8):
        ctxs = [_lib.Context(0) for _ in range(world)]
        try:
            stores = [separator.DimShardStore(c, keys, p, dp2scaf, scaf_len, so, si, 30) for c, p in zip(ctxs, separator.dim_partition(keys, world))]
            res, exchanged = separator.binning_tree_dims_local(stores)
            for rows_k, s2c_k in res:
                assert [_tree_key(r) for r in rows_k] == [_tree_key(r) for r in nodes] and np.array_equal(s2c_k, s2c)
            assert exchanged < 100 * (S + 40 * 1024)              # per level a byte per scaffold and a record per node
            del stores
        finally:
            for c in ctxs:
                c.close()
    # ... and with the SCAFFOLDS of the signature columns sharded over 2 and 8 ranks, the coverage columns by dimension
    # (abw_sep_stree_*): the same nodes and bins on every rank, at most 8 MB exchanged per rank and level
    for world in (2, 8):
        pairs = [(_lib.Context(0), _lib.Context(0)) for _ in range(world)]
        try:
            shards = separator.make_scaf_shards(pairs, keys, dp2scaf, scaf_len, so, si, 30)
            res, exchanged, levels = separator.binning_stree_local(shards)
            for rows_k, s2c_k in res:
                assert [_tree_key(r) for r in rows_k] == [_tree_key(r) for r in nodes] and np.array_equal(s2c_k, s2c)
            assert levels >= 10 and exchanged <= levels * (8 << 20)
            del shards
        finally:
            for a, b in pairs:
                a.close(); b.close()
    by_id = {r["id"]: r for r in nodes}
    assert nodes[0]["ndps"] == N and nodes[0]["split"] == 1
    # root winner from its definition (reference ...Specificity.cpp:6-76, .h:74-75)
    r0 = nodes[0]
    le = keys[r0["dim"] - 1] <= r0["key"]
    n1 = np.bincount(dp2scaf, weights=le, minlength=S).astype(np.int64)
    a1 = 2 * n1 > ndps
    size1 = int(le.sum())
    small1 = size1 < N - size1
    TP = int(n1[a1].sum()) if small1 else int((ndps - n1)[~a1].sum())
    TOT = int(ndps[a1].sum()) if small1 else int(ndps[~a1].sum())
    assert r0["sens"] == TP / TOT and r0["spec"] == TP / (size1 if small1 else N - size1)
    # well-formed tree: children partition their parent, ids in work-list order, every scaffold in exactly one bin
    for r in nodes:
        if r["split"]:
            assert r["n1"] + r["n2"] == r["ndps"] and r["n1"] >= 100 and r["n2"] >= 100
            assert by_id[r["c1"]]["ndps"] == r["n1"] and by_id[r["c2"]]["ndps"] == r["n2"]
            assert r["sens"] >= 0.8 and r["spec"] >= 0.8 and r["sens"] * r["spec"] >= 0.8
    bins = [r for r in nodes if not r["split"]]
    assert len(nodes) == 2 * len(bins) - 1
    assert s2c.min() > 0 and set(np.unique(s2c)) == {r["id"] for r in bins}
    sizes = np.bincount(s2c, weights=ndps, minlength=max(by_id) + 1)
    assert all(int(sizes[r["id"]]) == r["ndps"] for r in bins)
    order = np.argsort(s2c, kind="stable")
    cuts = np.flatnonzero(np.diff(s2c[order])) + 1
    agree = sum(int(np.bincount(g).max()) for g in np.split(np.asarray(genome)[order], cuts))
    assert agree / S > 0.99 and len(bins) >= G
    want = [_tree_key(r) for r in nodes]
    # ---- the same recursion with the scaffolds sharded over 2 and over 8 contexts
    for world, max_nodes in ((2, 1 << 30), (8, 255)):
        b = multi.scaffold_bounds(ndps, world)
        groups = [np.arange(b[k], b[k + 1]) for k in range(world)]
        parts = [separator.carve_shard(keys, dp2scaf, scaf_len, g, scg_scaf, scg_id) for g in groups]
        dv, do = multi.merge_dictionaries([p["keys"] for p in parts])
        ctxs = [_lib.Context(0) for _ in parts]
        try:
            stores = []
            for cc, p in zip(ctxs, parts):
                po, pi = separator.scg_csr(p["scg_scaf"], p["scg_id"], len(p["scaf_len"]))
                stores.append(separator.ShardedColumnStore(cc, p["keys"], p["dp2scaf"], p["scaf_len"], dv, do, po, pi, 30))
            del parts
            rows, s2c_parts = separator.binning_shards_local(stores, max_nodes=max_nodes)
            assert [_tree_key(r) for r in rows] == want[:len(rows)]
            if max_nodes >= len(nodes):
                got = np.concatenate(s2c_parts)
                assert np.array_equal(got, s2c)
        finally:
            for cc in ctxs:
                cc.close()
