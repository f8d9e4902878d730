"""Point-sharded k-means (abw_km_shard_*, SURVEY §8(e) second way) on ONE GPU: several contexts play the
shards, the test plays the collective (sums added in shard order, as abawaca_amd.multi does over RCCL)
and the result must equal the sharded CPU restatement bit for bit."""
import numpy as np
import pytest
import torch

import _oracle as orc
from abawaca_amd import _lib, kmeans, multi

pytestmark = pytest.mark.gpu


def _blobs(seed, N, d, sep=3.0):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 2, size=N)
    mu = rng.normal(0, 1, size=(2, d)) * sep
    return np.ascontiguousarray((mu[lab] + rng.normal(0, 1, size=(N, d))).astype(np.float32).T)


@pytest.mark.parametrize("N,d,R,iters,chunk,world", [(3000, 33, 5, 4, 64, 3), (5000, 189, 64, 3, 256, 2), (777, 7, 40, 2, 32, 4)])
def test_sharded_run_matches_sharded_oracle(N, d, R, iters, chunk, world):
    x = _blobs(N + d + world, N, d)
    bounds = multi.shard_bounds(N, world)
    dev = torch.device("cuda", 0)
    ctxs = [_lib.Context(0) for _ in range(world)]
    try:
        xs = [torch.from_numpy(np.ascontiguousarray(x[:, bounds[k]:bounds[k + 1]])).to(dev) for k in range(world)]
        runs = [kmeans.ShardRun(ctxs[k], xs[k], bounds[k + 1] - bounds[k], bounds[k], N, d, R, 555, chunk) for k in range(world)]

        def sync():
            for c in ctxs:
                c.synchronize()
            torch.cuda.synchronize()

        def combine(name, integer=False):
            sync()
            bufs = [getattr(r, name) for r in runs]
            tot = bufs[0].clone()
            for b in bufs[1:]:
                tot += b                                      # shard order
            for b in bufs:
                b.copy_(tot)
            sync()

        combine("init", integer=True)
        combine("moments")
        for r in runs:
            r.ready()
        asg = [torch.empty((R, bounds[k + 1] - bounds[k]), dtype=torch.uint8, device=dev) for k in range(world)]
        for it in range(iters + 1):
            last = it == iters
            for k, r in enumerate(runs):
                r.step(asg[k] if last else None)
            combine("sums")
            for r in runs:
                r.update(last)
        cents = [torch.empty((R, 2, d), dtype=torch.float32, device=dev) for _ in range(world)]
        inert = [torch.empty(R, dtype=torch.float64, device=dev) for _ in range(world)]
        for k, r in enumerate(runs):
            r.finish(cents[k], inert[k])
        sync()
        oa, oc, oi = orc.kmeans_sharded(x, R, 555, iters, chunk, bounds)
        for k in range(world):
            assert np.array_equal(cents[k].cpu().numpy().view(np.uint32), oc.view(np.uint32))     # every shard, identical bits
            assert np.array_equal(inert[k].cpu().numpy(), oi)
            assert np.array_equal(asg[k].cpu().numpy(), oa[:, bounds[k]:bounds[k + 1]])
    finally:
        for c in ctxs:
            c.close()


def test_one_shard_over_the_distributed_driver_is_the_plain_run():
    N, d, R, iters = 4096, 20, 8, 3
    x = _blobs(3, N, d)
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    try:
        t_x = torch.from_numpy(x).to(dev)
        a = torch.empty((R, N), dtype=torch.uint8, device=dev)
        c = torch.empty((R, 2, d), dtype=torch.float32, device=dev)
        i = torch.empty(R, dtype=torch.float64, device=dev)
        kmeans.kmeans_sharded_dev(ctx, t_x, 0, N, R, 9, iters, 128, None, a, c, i)
        ctx.synchronize()
        oa, oc, oi = orc.kmeans(x, R, 9, iters, 128)
        assert np.array_equal(a.cpu().numpy(), oa) and np.array_equal(c.cpu().numpy().view(np.uint32), oc.view(np.uint32))
        assert np.array_equal(i.cpu().numpy(), oi)
    finally:
        ctx.close()


def test_sum_ordered_is_the_rank_ordered_sum():
    """abw_km_sum_ordered = what multi.combine_ordered_sum does with torch adds: part 0 + part 1 + ... in that order"""
    ctx = _lib.Context(0)
    try:
        g = torch.Generator(device="cuda:0"); g.manual_seed(7)
        for world, n in ((2, 1), (5, 57344), (8, 114688 + 3)):
            parts = (torch.rand((world, n), device="cuda:0", generator=g, dtype=torch.float64) - 0.5) * 1e6
            want = parts[0].clone()
            for p in parts[1:]:
                want.add_(p)
            out = torch.empty(n, dtype=torch.float64, device="cuda:0")
            torch.cuda.synchronize()
            multi.sum_ordered(ctx, parts, out)
            ctx.synchronize()
            assert torch.equal(out.view(torch.int64), want.view(torch.int64))
            multi.sum_ordered(ctx, parts, parts[0])                     # in place, onto the first part
            ctx.synchronize()
            assert torch.equal(parts[0].view(torch.int64), want.view(torch.int64))
    finally:
        ctx.close()


def test_fast_combine_over_rccl_with_one_rank():
    """The RCCL path of multi.combine_ordered_sum (all_gather_into_tensor + one kernel on torch's stream) cannot meet a second
    GPU here; with a one-rank RCCL group its steps still run for real: the gathered buffer is the tensor itself."""
    import torch.distributed as dist
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    ctx = _lib.Context(0)
    try:
        ctx.use_torch_stream(True)
        t = torch.arange(1000, dtype=torch.float64, device="cuda:0") * 0.37
        want = t.clone()
        buf = torch.empty((1, t.numel()), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(buf, t.reshape(-1))
        multi.sum_ordered(ctx, buf, t)
        torch.cuda.synchronize()
        assert torch.equal(t, want)
        # and the function itself (one rank: nothing to add, returns the tensor)
        assert multi.combine_ordered_sum(t, dist, ctx) is t
    finally:
        ctx.use_torch_stream(False)
        ctx.close()
        dist.destroy_process_group()
