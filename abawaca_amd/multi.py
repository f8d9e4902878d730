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
