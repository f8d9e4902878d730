"""Multi-GPU sharding of the k-means restarts (SURVEY §8 row e: "restarts are spread across GPUs").

One process per GPU.  Restart r of the job always seeds TinyMT32 with base_seed + r, whichever rank
runs it, so a job gives the same restarts for any world size; a rank owns a contiguous range of
restart ids.  No data-path collective: every rank holds a replica of the points, the only exchange
is the gather of the per-restart inertia (R doubles per rank) that picks the winner, plus the
harness's barrier / MAX of the elapsed time.  The same code runs over RCCL ("nccl") on GPUs and over
"gloo" on CPUs (tests/test_multi_gloo.py)."""
from __future__ import annotations


def _active(dist) -> bool:
    return dist is not None and dist.is_initialized() and dist.get_world_size() > 1


def _coll_device(dist, device=None):
    """Where a tensor must live for a collective of the current backend: RCCL ("nccl") needs it on this
    rank's GPU, gloo takes host memory."""
    import torch
    if dist.get_backend() == "nccl":
        return device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def _all_reduce(t, op, dist):
    """In-place all-reduce of a tensor wherever it lives: over gloo a device tensor is staged through the host."""
    if t.is_cuda and dist.get_backend() != "nccl":
        h = t.cpu()
        dist.all_reduce(h, op=op)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op)
    return t


def restart_range(R_total: int, rank: int, world: int) -> tuple[int, int]:
    """(first restart id, number of restarts) of `rank` when R_total restarts are dealt out as evenly
    as possible, lower ranks taking the remainder."""
    if R_total < 0 or world < 1 or not 0 <= rank < world:
        raise ValueError("bad restart_range arguments")
    q, rem = divmod(R_total, world)
    first = rank * q + min(rank, rem)
    return first, q + (1 if rank < rem else 0)


def rank_seed(base_seed: int, first_restart: int) -> int:
    """base_seed argument for a rank whose first restart id is first_restart (abw_km_run seeds restart
    i of the call with base_seed + i)."""
    return (int(base_seed) + int(first_restart)) & 0xFFFFFFFF


def best_restart(inertia_local, first_restart: int, dist=None, device=None):
    """Global winner = smallest inertia, ties to the smaller restart id.  inertia_local: sequence of
    float64 for this rank's restarts.  Returns (restart id, inertia, owner rank)."""
    import torch
    loc = torch.as_tensor(inertia_local, dtype=torch.float64).flatten()
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        if loc.numel() == 0:
            raise ValueError("no restarts")
        i = int(torch.argmin(loc))          # first minimum
        return first_restart + i, float(loc[i]), 0
    world = dist.get_world_size()
    dev = _coll_device(dist, device if device is not None else (loc.device if loc.is_cuda else None))
    # ragged ranges: exchange (count, first id), then inertia padded with +inf
    meta = torch.tensor([loc.numel(), first_restart], dtype=torch.int64, device=dev)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    nmax = max(int(m[0]) for m in metas)
    pad = torch.full((nmax,), float("inf"), dtype=torch.float64, device=dev)
    pad[:loc.numel()] = loc.to(dev)
    allv = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(allv, pad)
    best = None
    for rk in range(world):
        n, f0 = int(metas[rk][0]), int(metas[rk][1])
        for i in range(n):
            cand = (float(allv[rk][i]), f0 + i, rk)
            if best is None or cand[:2] < best[:2]:
                best = cand
    if best is None:
        raise ValueError("no restarts")
    return best[1], best[0], best[2]


def max_elapsed(elapsed: float, dist=None, device=None) -> float:
    """MAX over ranks of a wall-clock interval (the bench contract)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed)
    import torch
    t = torch.tensor([elapsed], dtype=torch.float64, device=_coll_device(dist, device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


# ---- point-sharded k-means (SURVEY §8 row e, second way): the exchange between the library's phases ----

def shard_bounds(N: int, world: int) -> list[int]:
    """Contiguous point ranges, multiples of 32 points except the last (whole MFMA tiles per shard)."""
    per = -(-N // world)
    per = -(-per // 32) * 32
    b = [min(N, k * per) for k in range(world + 1)]
    if any(b[k + 1] - b[k] < 32 for k in range(world)):
        raise ValueError("every shard needs at least 32 points")
    return b


def combine_int_sum(t, dist=None):
    """In-place integer sum over the shards of an int32 tensor (initial centroids: bit patterns of the
    owner, zero elsewhere).  Exact and order-free."""
    if _active(dist):
        _all_reduce(t, dist.ReduceOp.SUM, dist)
    return t


_fast_combine = {}       # (backend, world, with a context, on the device) -> bool: the path every rank of this group takes, agreed once (see below)


def _fast_combine_agreed(t, dist, ctx) -> bool:
    """Whether ALL ranks can take the one-buffer all-gather + abw_km_sum_ordered.  Decided once per process group and
    collectively (an all-reduce MIN of each rank's own answer): ranks that disagreed would issue different collectives and
    hang.  Nothing is caught around the collectives themselves: a failure after one has been entered propagates."""
    import torch
    key = (dist.get_backend(), dist.get_world_size(), ctx is not None, bool(t.is_cuda))      # (the same on all ranks: one program)
    if key not in _fast_combine:
        mine = int(ctx is not None and t.is_cuda and dist.get_backend() == "nccl" and bool(getattr(ctx, "shares_torch_stream", False))
                   and hasattr(dist, "all_gather_into_tensor"))
        flag = torch.tensor([mine], dtype=torch.int32, device=t.device if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        _fast_combine[key] = bool(int(flag.item()))
    return _fast_combine[key]


def combine_ordered_sum(t, dist=None, ctx=None):
    """In-place fp64 sum over the shards, added in shard (rank) order on every rank, so that all ranks
    hold identical bits and the result does not depend on the collective's internal order: all-gather
    (a few hundred KB per rank, one direct exchange per peer over xGMI) + local adds.  With `ctx` (a Context whose
    stream is torch's current stream) on RCCL: one all-gather into one buffer and one kernel (abw_km_sum_ordered) --
    when every rank can (agreed once, _fast_combine_agreed)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    import torch
    world = dist.get_world_size()
    if _fast_combine_agreed(t, dist, ctx):
        buf = torch.empty((world, t.numel()), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, t.reshape(-1))
        sum_ordered(ctx, buf, t)
        return t
    src = t.cpu() if (t.is_cuda and dist.get_backend() != "nccl") else t
    parts = [torch.empty_like(src) for _ in range(world)]
    dist.all_gather(parts, src)
    acc = parts[0].clone()
    for p in parts[1:]:
        acc.add_(p)
    t.copy_(acc)
    return t


def sum_ordered(ctx, t_parts, t_out):
    """t_out[i] = ((t_parts[0][i] + t_parts[1][i]) + ...) -- fp64 device tensors, t_parts [n_parts, n] contiguous; one
    kernel on the context's stream (abw_km_sum_ordered)."""
    import ctypes as C
    from ._lib import lib
    n_parts, n = int(t_parts.shape[0]), int(t_out.numel())
    assert t_parts.is_contiguous() and t_out.is_contiguous() and t_parts.numel() == n_parts * n
    ctx.check(lib().abw_km_sum_ordered(ctx.handle, C.c_void_p(t_parts.data_ptr()), n_parts, n, C.c_void_p(t_out.data_ptr())))
    return t_out


# ---- scaffold-sharded separator (SURVEY §8 row e): shard layout, shared bin layout, the exchange ----

NARROW_T = 1024      # abw_sep.hip: a dimension whose keys all lie in [0, NARROW_T) uses the key itself as bin


def scaffold_bounds(scaf_ndps, world: int) -> list[int]:
    """Contiguous scaffold ranges with about the same number of dps each (whole scaffolds per shard)."""
    import numpy as np
    cum = np.concatenate([[0], np.cumsum(np.asarray(scaf_ndps, dtype=np.int64))])
    S, N = len(cum) - 1, int(cum[-1])
    b = [0]
    for k in range(1, world):
        b.append(max(b[-1], int(np.searchsorted(cum, k * N / world, side="left"))))
    b.append(S)
    if any(b[k + 1] <= b[k] for k in range(world)):
        raise ValueError("every shard needs at least one scaffold")
    return b


def global_dictionary(keys_local, dist=None, device=None):
    """Bin layout shared by all shards, from each shard's own [D, N_local] int32 keys: dimension d is wide when
    any shard holds a key outside [0, NARROW_T); its dictionary is the sorted union of the shards' distinct
    keys.  -> (dict_vals int32[...], dict_off int64[D+1]) for abw_sep_shard_load.  One MIN/MAX all-reduce of D
    values and one all-gather of the local distinct keys of the wide columns, once per run."""
    import numpy as np
    keys_local = np.asarray(keys_local)
    D = keys_local.shape[0]
    mn = keys_local.min(axis=1).astype(np.int64)
    mx = keys_local.max(axis=1).astype(np.int64)
    multi_rank = _active(dist)
    if multi_rank:
        import torch
        cdev = _coll_device(dist, device)
        tmn, tmx = torch.from_numpy(mn).to(cdev), torch.from_numpy(mx).to(cdev)
        dist.all_reduce(tmn, op=dist.ReduceOp.MIN)
        dist.all_reduce(tmx, op=dist.ReduceOp.MAX)
        mn, mx = tmn.cpu().numpy(), tmx.cpu().numpy()
    wide = (mn < 0) | (mx >= NARROW_T)
    uniq = [np.unique(keys_local[d]).astype(np.int32) if wide[d] else np.empty(0, np.int32) for d in range(D)]
    if multi_rank:
        import torch
        world = dist.get_world_size()
        flat = np.concatenate(uniq) if wide.any() else np.empty(0, np.int32)
        sizes = torch.tensor([len(u) for u in uniq], dtype=torch.int64, device=cdev)
        all_sizes = [torch.empty_like(sizes) for _ in range(world)]
        dist.all_gather(all_sizes, sizes)
        all_sizes = [s.cpu() for s in all_sizes]
        nmax = max(int(s.sum()) for s in all_sizes)
        pad = torch.zeros(max(nmax, 1), dtype=torch.int32)
        pad[:len(flat)] = torch.from_numpy(flat)
        pad = pad.to(cdev)
        parts = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(parts, pad)
        parts = [p.cpu() for p in parts]
        merged = []
        offs = [np.concatenate([[0], np.cumsum(s.numpy())]) for s in all_sizes]
        for d in range(D):
            if not wide[d]:
                merged.append(np.empty(0, np.int32))
                continue
            merged.append(np.unique(np.concatenate([parts[r].numpy()[offs[r][d]:offs[r][d + 1]] for r in range(world)])).astype(np.int32))
        uniq = merged
    dict_off = np.concatenate([[0], np.cumsum([len(u) for u in uniq])]).astype(np.int64)
    dict_vals = np.concatenate(uniq).astype(np.int32) if wide.any() else np.empty(0, np.int32)
    return dict_vals, dict_off


def merge_dictionaries(keys_per_shard):
    """global_dictionary for shards that live in one process (tests, single-node multi-context drivers)."""
    import numpy as np
    D = keys_per_shard[0].shape[0]
    mn = np.min([k.min(axis=1) for k in keys_per_shard], axis=0)
    mx = np.max([k.max(axis=1) for k in keys_per_shard], axis=0)
    wide = (mn < 0) | (mx >= NARROW_T)
    uniq = [np.unique(np.concatenate([k[d] for k in keys_per_shard])).astype(np.int32) if wide[d] else np.empty(0, np.int32)
            for d in range(D)]
    dict_off = np.concatenate([[0], np.cumsum([len(u) for u in uniq])]).astype(np.int64)
    dict_vals = np.concatenate(uniq).astype(np.int32) if wide.any() else np.empty(0, np.int32)
    return dict_vals, dict_off


def combine_sep(hist, scg_lo, scg_hi, dist=None):
    """The separator's exchange, in place: integer SUM of the histograms (int64 words; per node
    20 B x total bins: 3.7 MB for 179 narrow dimensions + 20 B per distinct coverage key), MIN / MAX of the
    per-SCG flip values.  All exact and order-free."""
    if not _active(dist):
        return
    _all_reduce(hist, dist.ReduceOp.SUM, dist)
    if scg_lo is not None:
        _all_reduce(scg_lo, dist.ReduceOp.MIN, dist)
        _all_reduce(scg_hi, dist.ReduceOp.MAX, dist)


def sum_counts(counts, dist=None, device=None):
    """Integer sum over the shards of a short list of counts (the dps each shard sent to cluster1 / cluster2)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [int(c) for c in counts]
    import torch
    t = torch.tensor([int(c) for c in counts], dtype=torch.int64, device=_coll_device(dist, device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(v) for v in t.tolist()]


# ---- signature build: scaffolds dealt to the ranks by length, no exchange (SURVEY §8 row e) ----

def scaffold_ranges_by_length(scaf_off, world: int) -> list[int]:
    """Contiguous scaffold ranges with about the same number of bases each: rank k builds the signatures of
    scaffolds [b[k], b[k+1]); segments are independent, so the ranks' rows simply concatenate in scaffold order."""
    import numpy as np
    off = np.asarray(scaf_off, dtype=np.int64)
    n, total = len(off) - 1, int(off[-1]) - int(off[0])
    b = [0]
    for k in range(1, world):
        b.append(min(n, max(b[-1], int(np.searchsorted(off, off[0] + k * total / world, side="left")))))
    b.append(n)
    return b
