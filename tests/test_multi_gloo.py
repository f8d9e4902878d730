"""World-size-2 (and 3, ragged) checks of the restart sharding over gloo on CPUs: the ranks' restarts
together are exactly the single-process job, and every rank agrees on the winner."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import _oracle as orc
from abawaca_amd import multi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _points():
    rng = np.random.default_rng(5)
    lab = rng.integers(0, 2, size=600)
    mu = rng.normal(0, 1, size=(2, 9)) * 2.0
    return np.ascontiguousarray((mu[lab] + rng.normal(0, 1, size=(600, 9))).astype(np.float32).T)


def _worker(rank, world, port, R_total, base_seed, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = _points()
        first, n = multi.restart_range(R_total, rank, world)
        if n:
            a, c, inert = orc.kmeans(x, n, multi.rank_seed(base_seed, first), 3, 64)
        else:
            a, c, inert = np.zeros((0, x.shape[1]), np.uint8), np.zeros((0, 2, x.shape[0]), np.float32), np.zeros(0)
        win = multi.best_restart(inert, first, dist)
        t = multi.max_elapsed(0.5 + rank, dist)
        dist.barrier()
        np.savez(os.path.join(out_dir, f"r{rank}.npz"), first=first, n=n, a=a, c=c, inert=inert, win=np.array(win, dtype=np.float64), t=t)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,R_total", [(2, 6), (3, 7), (2, 1)])
def test_restarts_spread_over_ranks(tmp_path, world, R_total):
    base_seed = 4242
    mp.spawn(_worker, args=(world, _free_port(), R_total, base_seed, str(tmp_path)), nprocs=world, join=True)
    x = _points()
    a1, c1, i1 = orc.kmeans(x, R_total, base_seed, 3, 64)
    covered = 0
    for rank in range(world):
        z = np.load(tmp_path / f"r{rank}.npz")
        first, n = int(z["first"]), int(z["n"])
        assert first == covered
        covered += n
        assert np.array_equal(z["a"], a1[first:first + n])
        assert np.array_equal(z["c"].view(np.uint32), c1[first:first + n].view(np.uint32))
        assert np.array_equal(z["inert"], i1[first:first + n])
        best = int(np.argmin(i1))
        assert int(z["win"][0]) == best and z["win"][1] == i1[best]
        assert first <= best < first + n if int(z["win"][2]) == rank else True
        assert z["t"] == 0.5 + (world - 1)                    # MAX over ranks
    assert covered == R_total


def test_restart_range_is_a_partition():
    for R in range(0, 20):
        for world in range(1, 9):
            got = [multi.restart_range(R, r, world) for r in range(world)]
            assert got[0][0] == 0 and sum(n for _, n in got) == R
            assert all(got[i][0] + got[i][1] == got[i + 1][0] for i in range(world - 1))
            assert max(n for _, n in got) - min(n for _, n in got) <= 1


def _combine_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        sums = torch.randn(257, dtype=torch.float64, generator=g) * 10.0 ** float(rank * 6)   # badly scaled: order matters
        init = torch.zeros(64, dtype=torch.int32)
        init[rank::world] = torch.arange(64, dtype=torch.int32)[rank::world] - 2 ** 31 + 5       # each word owned by one rank
        multi.combine_ordered_sum(sums, dist)
        multi.combine_int_sum(init, dist)
        torch.save(dict(sums=sums, init=init), os.path.join(out_dir, f"c{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_exchange_helpers_are_order_fixed_and_identical_on_every_rank(tmp_path):
    world = 3
    mp.spawn(_combine_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = []
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        parts.append(torch.randn(257, dtype=torch.float64, generator=g) * 10.0 ** float(rank * 6))
    want = (parts[0] + parts[1]) + parts[2]                   # shard order
    for rank in range(world):
        z = torch.load(tmp_path / f"c{rank}.pt")
        assert torch.equal(z["sums"].view(torch.int64), want.view(torch.int64))
        assert torch.equal(z["init"], torch.arange(64, dtype=torch.int32) - 2 ** 31 + 5)


def test_sharded_oracle_reduces_to_the_plain_one_and_stays_close():
    x = _points()
    a1, c1, i1 = orc.kmeans(x, 4, 77, 3, 64)
    a2, c2, i2 = orc.kmeans_sharded(x, 4, 77, 3, 64, [0, x.shape[1]])
    assert np.array_equal(a1, a2) and np.array_equal(c1.view(np.uint32), c2.view(np.uint32)) and np.array_equal(i1, i2)
    b = multi.shard_bounds(x.shape[1], 3)
    assert b[0] == 0 and b[-1] == x.shape[1] and all((b[k + 1] - b[k]) % 32 == 0 for k in range(2))
    a3, c3, i3 = orc.kmeans_sharded(x, 4, 77, 3, 64, b)
    assert (a3 != a1).mean() < 0.01                            # same algorithm, another summation order
    assert np.allclose(c3, c1, rtol=1e-4, atol=1e-4) and np.allclose(i3, i1, rtol=1e-6)


# ---- scaffold-sharded separator: shared bin layout + the exchange, with a numpy model of the histograms ----

def _np_local_hist(keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, dict_vals, dict_off):
    """What a shard contributes: per dimension size / tp1 / tot / bp histograms over the shared
    bins and per-SCG min / max flip value.  Plain loops: test model of abw_sep_shard_accum's output."""
    D, N = keys.shape
    S = len(scaf_len)
    T = [int(dict_off[d + 1] - dict_off[d]) or multi.NARROW_T for d in range(D)]
    Tmax = max(T)
    hist = np.zeros((D, 4, Tmax), dtype=np.int64)
    lo = np.full((D, max(n_scg, 1)), np.iinfo(np.int32).max, dtype=np.int32)
    hi = np.full((D, max(n_scg, 1)), -1, dtype=np.int32)
    ndps = np.bincount(dp2scaf, minlength=S)
    first = np.concatenate([[0], np.cumsum(ndps)[:-1]])
    genes = [[] for _ in range(S)]
    for s, g in zip(scg_scaf, scg_id):
        genes[int(s)].append(int(g))
    for d in range(D):
        col = keys[d]
        if dict_off[d + 1] > dict_off[d]:
            col = np.searchsorted(dict_vals[dict_off[d]:dict_off[d + 1]], col)
        for s in range(S):
            v = np.sort(col[first[s]:first[s] + ndps[s]])
            ts = v[ndps[s] // 2]
            np.add.at(hist[d, 0], v, 1)
            np.add.at(hist[d, 1], np.maximum(v, ts), 1)
            hist[d, 2, ts] += ndps[s]
            hist[d, 3, ts] += scaf_len[s]
            for g in genes[s]:
                lo[d, g] = min(lo[d, g], ts)
                hi[d, g] = max(hi[d, g], ts)
    return hist, lo, hi, T


def _np_eval(hist, lo, hi, T, n_scg, min_ndps=100):
    """Per-dimension winner from the combined histograms (reference rules: ...Specificity.cpp:146-161, .h:25-30,
    ClusterQuality.cpp:96-136) -> list of (found, bin, sens, spec)."""
    out = []
    for d in range(hist.shape[0]):
        size1, tp1, tot1, bp1 = (np.cumsum(hist[d, k, :T[d]]) for k in range(4))
        n, bp = int(size1[-1]), int(bp1[-1])
        l, h = lo[d][lo[d] != np.iinfo(np.int32).max], hi[d][lo[d] != np.iinfo(np.int32).max]
        best = (0, -1, 0.0, 0.0)
        bp_ = -1.0
        for t in np.flatnonzero(hist[d, 0, :T[d]]):
            s1, s2 = int(size1[t]), n - int(size1[t])
            if s1 < min_ndps or s2 < min_ndps:
                continue
            small1 = s1 < s2
            TP = int(tp1[t]) if small1 else n - (s1 + int(tot1[t]) - int(tp1[t]))
            TOT = int(tot1[t]) if small1 else n - int(tot1[t])
            SZ = s1 if small1 else s2
            if TOT == 0:
                continue
            sens, spec = TP / TOT, TP / SZ
            if not sens * spec > bp_:
                continue
            G1, G2, both = int((l <= t).sum()), int((h > t).sum()), int(((l <= t) & (h > t)).sum())
            if G1 == 0 or G2 == 0:
                ok = bp1[t] >= 500000 and bp - bp1[t] >= 500000
            else:
                ok = not (both / min(G1, G2) < 0.2)
            if ok:
                best, bp_ = (1, int(t), sens, spec), sens * spec
        out.append(best)
    return out


def _sep_tables():
    g = np.load(os.path.join(ROOT, "tests", "golden", "sep_g3.npz"))
    dims = np.r_[0:6, 40:44, 179:182]                         # a few signature columns + the coverage columns
    return np.ascontiguousarray(g["milli"][dims]), g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"])


def _sep_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from abawaca_amd import separator
        keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg = _sep_tables()
        S = len(scaf_len)
        mine = np.arange(rank, S, world)                      # scaffolds dealt round-robin
        p = separator.carve_shard(keys, dp2scaf, scaf_len, mine, scg_scaf, scg_id)
        dv, do = multi.global_dictionary(p["keys"], dist)
        hist, lo, hi, T = _np_local_hist(p["keys"], p["dp2scaf"], p["scaf_len"], p["scg_scaf"], p["scg_id"], n_scg, dv, do)
        th, tl, thi = torch.from_numpy(hist), torch.from_numpy(lo), torch.from_numpy(hi)
        multi.combine_sep(th, tl, thi, dist)
        tot = multi.sum_counts([len(mine), int(p["keys"].shape[1])], dist)
        per = _np_eval(th.numpy(), tl.numpy(), thi.numpy(), T, n_scg)
        keyof = [(int(dv[do[d] + b]) if do[d + 1] > do[d] else b) if f else -1 for d, (f, b, _, _) in enumerate(per)]
        np.savez(os.path.join(out_dir, f"s{rank}.npz"), dv=dv, do=do, found=[x[0] for x in per], key=keyof,
                 sens=[x[2] for x in per], spec=[x[3] for x in per], tot=tot)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_separator_histograms_add_up_to_the_reference_sweep(tmp_path, world):
    mp.spawn(_sep_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg = _sep_tables()
    S = len(scaf_len)
    off, ids = orc.scg_csr(scg_scaf, scg_id, S)
    ndps = np.bincount(dp2scaf, minlength=S).astype(np.int32)
    res, per, raw, side = orc.separate(keys, dp2scaf, ndps, scaf_len, off, ids, n_scg, np.arange(keys.shape[1], dtype=np.int32), 100)
    dv1, do1 = multi.merge_dictionaries([keys])
    assert per["found"].sum() > 0
    for rank in range(world):
        z = np.load(tmp_path / f"s{rank}.npz")
        assert np.array_equal(z["dv"], dv1) and np.array_equal(z["do"], do1)        # every rank built the same layout
        assert list(z["tot"]) == [S, keys.shape[1]]
        assert np.array_equal(z["found"], per["found"])
        f = per["found"] == 1
        assert np.array_equal(z["key"][f], per["key"][f])
        assert np.array_equal(z["sens"][f], per["sens"][f]) and np.array_equal(z["spec"][f], per["spec"][f])


# ---- scaffold-sharded LEVEL exchange (abw_sep_stree_*): narrow columns by scaffold range, wide columns by dimension ----

def _stree_worker(rank, world, port, out_dir):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from abawaca_amd import separator
        keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg = _sep_tables()
        D, S = keys.shape[0], len(scaf_len)
        wide = np.array([keys[d].max() >= 1024 for d in range(D)])
        n_cols, w_cols = np.flatnonzero(~wide), np.flatnonzero(wide)
        b = separator.scaffold_ranges(np.bincount(dp2scaf, minlength=S), world)
        # narrow columns of my scaffold range: local histograms (bin = key), summed over the ranks; SCG lists MIN / MAX
        p = separator.carve_shard(keys[n_cols], dp2scaf, scaf_len, np.arange(b[rank], b[rank + 1]), scg_scaf, scg_id)
        none = np.zeros(len(n_cols) + 1, dtype=np.int64)
        hist, lo, hi, Tn = _np_local_hist(p["keys"], p["dp2scaf"], p["scaf_len"], p["scg_scaf"], p["scg_id"], n_scg, np.zeros(0, np.int32), none)
        th, tl, thi = torch.from_numpy(hist), torch.from_numpy(lo), torch.from_numpy(hi)
        multi.combine_sep(th, tl, thi, dist)
        per_n = _np_eval(th.numpy(), tl.numpy(), thi.numpy(), Tn, n_scg)
        # my wide columns over ALL scaffolds: nothing to exchange
        mine_w = w_cols[rank::world]
        per_w, dvw, dow = [], None, None
        if len(mine_w):
            dvw, dow = multi.merge_dictionaries([keys[mine_w]])
            hw, lw, hiw, Tw = _np_local_hist(keys[mine_w], dp2scaf, scaf_len, scg_scaf, scg_id, n_scg, dvw, dow)
            per_w = _np_eval(hw, lw, hiw, Tw, n_scg)

        def best_of(per, cols, keyof):
            r = np.zeros(1, dtype=separator._WIN_DTYPE)
            r[0] = (0, -1, -1, 0, 0.0, 0.0, -1, 0)
            for i, (f, bn, sens, spec) in enumerate(per):
                if f and (not r["found"][0] or sens * spec > r["sens"][0] * r["spec"][0]):      # ascending columns: ties keep the smaller
                    legal = sens >= 0.8 and spec >= 0.8 and sens + spec >= 1.6 and sens * spec >= 0.8
                    r[0] = (1, i, keyof(i, bn), int(legal), sens, spec, int(cols[i]) + 1, 0)
            return r
        best_n = best_of(per_n, n_cols, lambda i, bn: bn)
        best_w = best_of(per_w, mine_w, lambda i, bn: int(dvw[dow[i] + bn]))
        mine = separator.stree_mine(best_n, best_w)
        mt = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
        parts = [torch.zeros_like(mt) for _ in range(world)]
        dist.all_gather(parts, mt)
        winners, owner = separator.merge_winners([q.numpy().view(separator._WIN_DTYPE) for q in parts])
        np.savez(os.path.join(out_dir, f"t{rank}.npz"), win=winners.view(np.uint8), owner=owner, narrow_best=best_n.view(np.uint8))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_scaffold_sharded_level_exchange_and_merge_find_the_reference_winner(tmp_path, world):
    """World-size 2 / 3 over gloo: the narrow columns' histograms of contiguous scaffold ranges summed over the ranks, every rank's
    own coverage columns over all scaffolds, each rank's candidate = the better of the two (separator.stree_mine), merged over
    the ranks (abw_sep_tree_merge) -> the node's winner as the oracle's sweep over the whole table finds it, on every rank."""
    from abawaca_amd import separator
    mp.spawn(_stree_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    keys, dp2scaf, scaf_len, scg_scaf, scg_id, n_scg = _sep_tables()
    S = len(scaf_len)
    off, ids = orc.scg_csr(scg_scaf, scg_id, S)
    ndps = np.bincount(dp2scaf, minlength=S).astype(np.int32)
    res, per, raw, side = orc.separate(keys, dp2scaf, ndps, scaf_len, off, ids, n_scg, np.arange(keys.shape[1], dtype=np.int32), 100)
    assert res.dimension > 0
    first = None
    for rank in range(world):
        z = np.load(tmp_path / f"t{rank}.npz")
        w = z["win"].view(separator._WIN_DTYPE)[0]
        assert (int(w["gdim"]), int(w["key"]), float(w["sens"]), float(w["spec"]), int(w["legal"])) == \
               (res.dimension, res.key, res.sensitivity, res.specificity, res.legal)
        first = z if first is None else first
        assert np.array_equal(z["win"], first["win"]) and np.array_equal(z["owner"], first["owner"])
        assert np.array_equal(z["narrow_best"], first["narrow_best"])                       # identical on every rank


def test_scaffold_bounds_balance_dps():
    rng = np.random.default_rng(0)
    ndps = rng.integers(2, 9, size=1000)
    for world in (1, 2, 3, 8):
        b = multi.scaffold_bounds(ndps, world)
        assert b[0] == 0 and b[-1] == 1000 and all(b[k] < b[k + 1] for k in range(world))
        per = [ndps[b[k]:b[k + 1]].sum() for k in range(world)]
        assert max(per) - min(per) <= 16


def test_signature_ranges_cover_every_scaffold_once_and_rows_concatenate():
    """Signature sharding has no exchange: the oracle on each rank's scaffold range, concatenated, is the oracle on
    the whole assembly."""
    from abawaca_amd import synth
    asm = synth.make_assembly(31, 40, 3, 1, frac_n_run=0.3)
    blob = np.concatenate(asm.seqs)
    off = np.zeros(len(asm.seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in asm.seqs])
    whole = orc.build_signatures(blob, off)
    for world in (1, 2, 3, 8):
        b = multi.scaffold_ranges_by_length(off, world)
        assert b[0] == 0 and b[-1] == len(asm.seqs) and all(b[k] <= b[k + 1] for k in range(world))
        parts = []
        for k in range(world):
            if b[k] == b[k + 1]:
                continue
            lo, hi = int(off[b[k]]), int(off[b[k + 1]])
            seg_scaf, st, en, dims = orc.build_signatures(blob[lo:hi], off[b[k]:b[k + 1] + 1] - off[b[k]])
            parts.append((seg_scaf + b[k], st, en, dims))
        assert np.array_equal(np.concatenate([p[0] for p in parts]), whole[0])
        assert np.array_equal(np.concatenate([p[1] for p in parts]), whole[1]) and np.array_equal(np.concatenate([p[2] for p in parts]), whole[2])
        assert np.array_equal(np.concatenate([p[3] for p in parts]).view(np.uint64), whole[3].view(np.uint64))
        sizes = [int(off[b[k + 1]] - off[b[k]]) for k in range(world)]
        assert max(sizes) - min(sizes) <= 2 * max(len(s) for s in asm.seqs)
