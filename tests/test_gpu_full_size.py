"""The hot path at BASELINE.json's full sizes (configs[1]: 200 000 scaffolds x 189 columns, 64 restarts; configs[3]:
10 kbp scaffolds, here 300 Mbp of them), where the CPU oracle would take too long: size-independent properties.

  separator : the winner's sensitivity / specificity recomputed from their definition (reference
              src/ClusterSeparatorBySensitivitySpecificity.cpp:6-76, .h:74-75) with numpy at the winning threshold are
              the doubles returned; children partition the node; 2 and 5 scaffold shards give the unsharded answer
  k-means   : restarts are independent of how they are batched (64 at once == 2 x 32 with shifted seeds, bit for bit);
              every point sits with the nearer returned centroid (float64 check, near-ties aside); inertia == direct sum
  signature : counts of every k sum to the number of valid windows computed independently; every segment's
              dimensions sum to 4; segments tile every scaffold's non-N bases in windows of [2000, 4000)
  .lrn      : the table written as abawaca.lrn rows (abw_lrn_emit) and as the rows of clusters (abw_lrn_emit_rows, all
              the clusters of a partition in one call) parses back (abw_lrn_parse) to the same keys; the row offsets cut
              the text at line starts; every row has the column count; "%lf" of a 3-decimal value is "%.3lf" + "000"
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the bench's generators are the BASELINE configs)
from abawaca_amd import _lib, kmeans, lrn, multi, separator, signature  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def table():
    import torch
    dev = torch.device("cuda", 0)
    keys_t, dp2scaf, scaf_len, genome, G = bench.make_points(20241, 200_000, 10, dev)
    return dict(keys_t=keys_t, keys=keys_t.cpu().numpy(), dp2scaf=dp2scaf, scaf_len=scaf_len, genome=genome)


def test_separator_root_of_configs1(table):
    keys, dp2scaf, scaf_len = table["keys"], table["dp2scaf"], table["scaf_len"]
    D, N = keys.shape
    S = len(scaf_len)
    ctx = _lib.Context(0)
    try:
        store = separator.ColumnStore(ctx, keys, dp2scaf, scaf_len)
        sep = separator.ClusterSeparatorHip(store)
        assert sep.separate()
        r = sep._res
        assert r.ndps1 + r.ndps2 == N and r.legal == 1
        # the winner from its definition: cluster1 = {v <= t}, a scaffold is assigned to cluster1 iff 2*n1_s > n_s
        col = keys[r.dimension - 1]
        le = col <= r.key
        n1 = np.bincount(dp2scaf, weights=le, minlength=S).astype(np.int64)
        ns = store.scaf_ndps.astype(np.int64)
        a1 = 2 * n1 > ns
        size1, size2 = int(le.sum()), N - int(le.sum())
        tp1, tot1 = int(n1[a1].sum()), int(ns[a1].sum())
        tp2, tot2 = int((ns - n1)[~a1].sum()), int(ns[~a1].sum())
        small1 = size1 < size2
        TP, TOT, SZ = (tp1, tot1, size1) if small1 else (tp2, tot2, size2)
        assert r.sensitivity == TP / TOT and r.specificity == TP / SZ            # bit-exact doubles
        # per-dimension winners: no dimension beats the global one, ties go to the smaller dimension
        per = sep.per_dimension()
        prod = np.where(per["found"] == 1, per["sens"] * per["spec"], -1.0)
        best = int(np.argmax(prod))
        assert best + 1 == r.dimension and prod[best] == r.sensitivity * r.specificity
        # final assignment (ref ClusterSeparator.cpp:94-122): cluster1 = the smaller raw side, 2*n >= total
        c1 = n1 if size1 <= size2 else ns - n1
        to1 = (c1 > 0) & (2 * c1 >= ns)
        assert np.array_equal(sep._side, np.where(to1, 1, 2)) and r.ndps1 == int(ns[to1].sum())
        whole_per, whole_res, whole_side = per.copy(), (r.split, r.dimension, r.key, r.sensitivity, r.specificity, r.ndps1, r.ndps2), sep._side.copy()
    finally:
        ctx.close()
    # the same node from 2 and 5 scaffold shards
    for world in (2, 5):
        b = multi.scaffold_bounds(np.bincount(dp2scaf, minlength=S), world)
        groups = [np.arange(b[k], b[k + 1]) for k in range(world)]
        parts = [separator.carve_shard(keys, dp2scaf, scaf_len, g) for g in groups]
        dv, do = multi.merge_dictionaries([p["keys"] for p in parts])
        ctxs = [_lib.Context(0) for _ in parts]
        try:
            stores = [separator.ShardedColumnStore(c, p["keys"], p["dp2scaf"], p["scaf_len"], dv, do) for c, p in zip(ctxs, parts)]
            nodes = separator.separate_shards_local(stores, [None] * world)
            for nd, g in zip(nodes, groups):
                o = nd.out
                assert (o.split, o.dimension, o.key, o.sensitivity, o.specificity, o.ndps1, o.ndps2) == whole_res
                assert np.array_equal(nd.per, whole_per) and np.array_equal(nd.side, whole_side[g])
        finally:
            for c in ctxs:
                c.close()


def test_kmeans_64_restarts_of_configs1(table):
    import torch
    keys_t = table["keys_t"]
    D, N = keys_t.shape
    x_t = bench.standardise(keys_t)
    torch.cuda.synchronize()          # the library runs on its own stream: x_t must be complete before it reads it
    dev = x_t.device
    ctx = _lib.Context(0)
    try:
        R, iters, chunk = 64, 6, kmeans.auto_chunk(ctx, N, 64)
        out = {}
        for tag, r, seed in (("all", 64, 500), ("lo", 32, 500), ("hi", 32, 532)):
            a = torch.empty((r, N), dtype=torch.uint8, device=dev)
            c = torch.empty((r, 2, D), dtype=torch.float32, device=dev)
            i = torch.empty(r, dtype=torch.float64, device=dev)
            kmeans.kmeans_restarts_dev(ctx, x_t, N, D, r, seed, iters, chunk, a, c, i)
            ctx.synchronize()
            out[tag] = (a, c, i)
        # restart r is the same whatever batch it runs in
        assert torch.equal(out["all"][2], torch.cat([out["lo"][2], out["hi"][2]]))
        assert torch.equal(out["all"][1].view(torch.int32), torch.cat([out["lo"][1], out["hi"][1]]).view(torch.int32))
        assert torch.equal(out["all"][0], torch.cat([out["lo"][0], out["hi"][0]]))
        a, c, inert = out["all"]
        best = int(torch.argmin(inert))
        xb = x_t.to(torch.float64)                                          # [D, N]
        cb = c[best].to(torch.float64)                                      # [2, D]
        d0 = ((xb - cb[0][:, None]) ** 2).sum(0)
        d1 = ((xb - cb[1][:, None]) ** 2).sum(0)
        mine = a[best].to(torch.bool)
        wrong = (d1 < d0) != mine
        near_tie = (d0 - d1).abs() <= 1e-4 * (d0 + d1)
        assert int((wrong & ~near_tie).sum()) == 0 and int(wrong.sum()) <= N // 10000       # only documented near-tie flips
        direct = float(torch.where(mine, d1, d0).sum())
        assert abs(direct - float(inert[best])) / direct < 1e-6
        assert 0 < int(mine.sum()) < N
    finally:
        ctx.close()


def test_signature_300mbp_of_configs3():
    import torch
    blob, off = bench.make_assembly_blob(20244, 300)
    total = len(blob)
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    try:
        t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-total) % 16 + 16, np.uint8)])).to(dev)
        t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        cap = total // 2000 + len(off) + 16
        t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
        t_st = torch.empty(cap, dtype=torch.int64, device=dev)
        t_en = torch.empty(cap, dtype=torch.int64, device=dev)
        n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
        t_dims = torch.empty((n, 180), dtype=torch.float64, device=dev)
        t_counts = torch.empty((n, 340), dtype=torch.int32, device=dev)
        signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims, t_counts=t_counts)
        ctx.synchronize()
        scaf = t_scaf[:n].cpu().numpy().astype(np.int64)
        st, en = t_st[:n].cpu().numpy(), t_en[:n].cpu().numpy()
        # segments: per scaffold max(1, nonN // 2000) of them, consecutive, starting at 1, each with nbps non-N bases
        isN = blob == ord("N")
        nonN_scaf = np.add.reduceat((~isN).astype(np.int64), off[:-1].astype(np.int64))
        nseg = np.maximum(1, nonN_scaf // 2000)
        assert n == int(nseg.sum()) and np.array_equal(np.bincount(scaf, minlength=len(off) - 1), nseg)
        first = np.concatenate([[0], np.cumsum(nseg)[:-1]])
        assert np.all(st[first] == 1) and np.all(st[1:][scaf[1:] == scaf[:-1]] == en[:-1][scaf[1:] == scaf[:-1]] + 1)
        cumN = np.concatenate([[0], np.cumsum(~isN)])
        b = off[scaf].astype(np.int64) + st - 1
        e = off[scaf].astype(np.int64) + en
        seg_nonN = cumN[e] - cumN[b]
        assert np.array_equal(seg_nonN, (nonN_scaf // nseg)[scaf])
        assert np.all(~isN[e - 1])                                          # a segment ends on the base that fills it
        # counts: valid windows of length k inside a segment = sum over its maximal ACGT runs of max(0, len - k + 1);
        # this assembly has only ACGT and N, so the runs are the non-N runs
        counts = t_counts.cpu().numpy().astype(np.int64)
        tot = np.stack([counts[:, o:o + w].sum(1) for o, w in ((0, 4), (4, 16), (20, 64), (84, 256))], axis=1)
        assert np.array_equal(tot[:, 0], seg_nonN)
        clean = (cumN[e] - cumN[b]) == (e - b)                              # segments without any N: one run
        for k in range(1, 5):
            assert np.array_equal(tot[clean, k - 1], (e - b)[clean] - k + 1)
        assert (~clean).sum() > 0 and np.all(tot[~clean, 3] < (e - b)[~clean] - 3)
        dims = t_dims.cpu().numpy()
        assert np.allclose(dims.sum(1), 4.0, atol=1e-12)
    finally:
        ctx.close()


def test_lrn_writers_round_trip_at_configs1(table):
    import torch
    keys = table["keys"]                                         # [189][N] thousandths
    D, N = keys.shape
    ctx = _lib.Context(0)
    try:
        head = (f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\tc{j}" for j in range(D)) + "\n").encode()
        # build writer: rows 1..N in order
        rows = np.ascontiguousarray(keys.T.astype(np.int32))
        text = lrn.lrn_emit(ctx, rows, 1)
        assert int((text == 10).sum()) == N and int((text == 9).sum()) == N * D
        dp, k2, _ = lrn.lrn_parse(ctx, np.concatenate([np.frombuffer(head, np.uint8), text]))
        assert np.array_equal(dp, np.arange(1, N + 1)) and np.array_equal(k2, keys)
        # cluster writer: a random partition into 64 clusters, all of them in one call; values = key / 1000.0 as the
        # reference's atof holds them
        rng = np.random.default_rng(1)
        perm = rng.permutation(N).astype(np.int64) + 1
        cuts = np.sort(rng.choice(np.arange(1, N), size=63, replace=False))
        first = np.concatenate([[0], cuts, [N]])
        t_vals = torch.from_numpy(keys.astype(np.float64) / 1000.0).cuda()
        t_dps = torch.from_numpy(perm).cuda()
        t_off = torch.zeros(N + 1, dtype=torch.int64, device="cuda")
        need = lrn.lrn_emit_rows_dev(ctx, t_vals, t_dps, None)
        t_text = torch.empty(need, dtype=torch.uint8, device="cuda")
        assert lrn.lrn_emit_rows_dev(ctx, t_vals, t_dps, t_text, t_off) == need
        ctext, off = t_text.cpu().numpy(), t_off.cpu().numpy()
        assert off[0] == 0 and off[-1] == need and np.all(np.diff(off) > 0)
        assert np.all(ctext[off[1:] - 1] == 10) and int((ctext == 10).sum()) == N      # every row ends where the next starts
        assert int((ctext == 9).sum()) == N * D
        # all clusters concatenated are one table with shuffled keys: the loader reads it back (6 decimals, the last three "000")
        dp3, k3, _ = lrn.lrn_parse(ctx, np.concatenate([np.frombuffer(head, np.uint8), ctext]))
        assert np.array_equal(dp3, perm) and np.array_equal(k3, keys[:, perm - 1])
        # one cluster cut out by its offsets == the same cluster written alone
        c = 17
        lo, hi = int(first[c]), int(first[c + 1])
        alone = torch.empty(int(off[hi] - off[lo]), dtype=torch.uint8, device="cuda")
        assert lrn.lrn_emit_rows_dev(ctx, t_vals, t_dps[lo:hi], alone) == alone.numel()
        assert np.array_equal(alone.cpu().numpy(), ctext[off[lo]:off[hi]])
        # "%lf" of a 3-decimal value is its "%.3lf" text followed by "000": compare the first rows of both writers
        r0 = bytes(ctext[off[0]:off[1]]).rstrip(b"\n").split(b"\t")
        b0 = (b"\n" + bytes(text)).split(b"\n" + str(int(perm[0])).encode() + b"\t", 1)[1].split(b"\n", 1)[0].split(b"\t")
        assert r0[0] == str(int(perm[0])).encode() and [x + b"000" for x in b0] == r0[1:]
    finally:
        ctx.close()
