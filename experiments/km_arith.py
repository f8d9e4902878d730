"""K-means job of configs[1]'s shape in both arithmetics, for rocprofv3 (kernel trace or --pmc): two jobs each.
usage: rocprofv3 --pmc FETCH_SIZE ... -- python3 experiments/km_arith.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from abawaca_amd import _lib, kmeans

dev = torch.device("cuda", 0)
keys_t, _dp2scaf, _len, _genome, _G = bench.make_points(20241, 200_000, 10, dev, 100)
x = bench.standardise(keys_t)
torch.cuda.synchronize()
D, N = x.shape
R, iters = 64, 20
ctx = _lib.Context(0)
chunk = kmeans.auto_chunk(ctx, N, R)
inert = torch.empty(R, dtype=torch.float64, device=dev)
for arith in (kmeans.ARITH_F32, kmeans.ARITH_BF16X3):
    kmeans.set_arith(ctx, arith)
    for _ in range(2):
        kmeans.kmeans_restarts_dev(ctx, x, N, D, R, 20240, iters, chunk, None, None, inert)
        ctx.synchronize()
    print(["f32", "bf16x3"][arith], "best inertia", float(inert.min()))
ctx.close()
