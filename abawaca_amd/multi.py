"""Multi-GPU sharding of the k-means restarts (SURVEY §8 row e: "restarts are spread across GPUs").

One process per GPU.  Restart r of the job always seeds TinyMT32 with base_seed + r, whichever rank
runs it, so a job gives the same restarts for any world size; a rank owns a contiguous range of
restart ids.  No data-path collective: every rank holds a replica of the points, the only exchange
is the gather of the per-restart inertia (R doubles per rank) that picks the winner, plus the
harness's barrier / MAX of the elapsed time.  The same code runs over RCCL ("nccl") on GPUs and over
"gloo" on CPUs (tests/test_multi_gloo.py)."""
from __future__ import annotations


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
    dev = device if device is not None else loc.device
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
    t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
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
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def combine_ordered_sum(t, dist=None):
    """In-place fp64 sum over the shards, added in shard (rank) order on every rank, so that all ranks
    hold identical bits and the result does not depend on the collective's internal order: all-gather
    (a few hundred KB per rank, one direct exchange per peer over xGMI) + local adds."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return t
    import torch
    parts = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t)
    t.copy_(parts[0])
    for p in parts[1:]:
        t.add_(p)
    return t
