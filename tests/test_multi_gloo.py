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
