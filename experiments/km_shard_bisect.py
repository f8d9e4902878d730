"""bisect: point-sharded vs unsharded k-means at growing sizes (2 shards on one device)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abawaca_amd import _lib, kmeans, multi

dev = torch.device("cuda", 0)
def run(N, d, R, chunk, world=2, iters=2):
    g = torch.Generator(device=dev); g.manual_seed(1)
    x = torch.randn((d, N), device=dev, generator=g)
    x[:, : N // 3] += 1.5
    torch.cuda.synchronize()
    ctx = _lib.Context(0)
    a = torch.empty((R, N), dtype=torch.uint8, device=dev); i = torch.empty(R, dtype=torch.float64, device=dev)
    kmeans.kmeans_restarts_dev(ctx, x, N, d, R, 7, iters, chunk, a, None, i); ctx.synchronize()
    b = multi.shard_bounds(N, world)
    ctxs = [_lib.Context(0) for _ in range(world)]
    xs = [x[:, b[k]:b[k + 1]].contiguous() for k in range(world)]
    torch.cuda.synchronize()          # the library runs on its own streams: the slices must be complete before it reads them
    runs = [kmeans.ShardRun(ctxs[k], xs[k], b[k + 1] - b[k], b[k], N, d, R, 7, chunk) for k in range(world)]
    def sync():
        for c in ctxs: c.synchronize()
        torch.cuda.synchronize()
    def combine(name):
        sync()
        bufs = [getattr(r, name) for r in runs]
        tot = bufs[0].clone()
        for t in bufs[1:]: tot += t
        for t in bufs: t.copy_(tot)
        sync()
    combine("init"); combine("moments")
    for r in runs: r.ready()
    for it in range(iters + 1):
        for r in runs: r.step(None)
        combine("sums")
        for r in runs: r.update(it == iters)
    inert = [torch.empty(R, dtype=torch.float64, device=dev) for _ in range(world)]
    for k, r in enumerate(runs): r.finish(None, inert[k])
    sync()
    rel = float(((inert[0] - i).abs() / i).max())
    print(f"N={N} d={d} R={R} chunk={chunk} world={world}: max rel inertia diff {rel:.3e}", flush=True)
    for c in ctxs: c.close()
    ctx.close()

run(5000, 189, 64, 256)
run(200_000, 189, 64, 1024)
run(200_000, 199, 256, 1024)
run(200_000, 199, 128, 1024)
run(200_000, 199, 96, 1024)
run(2_000_000, 199, 64, 1024)
run(4_000_000, 199, 256, 1024)
