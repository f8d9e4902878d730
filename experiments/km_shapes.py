"""km_step at the shapes `bench.py --gpus 8` gives one GPU of configs[2] (d = 199):
   restarts partition: 4.0 M points x 32 restarts;   points partition: 0.5 M points x 256 restarts
(+ the one-GPU shapes).  Prints ms per launch and the fraction of the fp32 MFMA peak at 4(d+1) flop/unit."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abawaca_amd import _lib, kmeans
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
g = torch.Generator(device=dev); g.manual_seed(1)
for N, d, R in [(3_999_008, 199, 32), (499_904, 199, 256), (3_999_008, 199, 256), (1_999_520, 199, 64), (999_776, 199, 128), (799_859, 189, 64), (799_859, 189, 32)]:
    x = torch.randn(d, N, device=dev, generator=g)
    inert = torch.empty(R, dtype=torch.float64, device=dev)
    torch.cuda.synchronize()
    chunk = kmeans.auto_chunk(ctx, N, R)
    kmeans.kmeans_restarts_dev(ctx, x, N, d, R, 5, 2, chunk, None, None, inert); ctx.synchronize()
    ctx.prof_reset(); ctx.prof_enable(True)
    kmeans.kmeans_restarts_dev(ctx, x, N, d, R, 5, 6, chunk, None, None, inert); ctx.synchronize()
    n, ms = ctx.prof_read("km_step"); nu, msu = ctx.prof_read("km_update")
    ctx.prof_enable(False)
    tf = N * R * 4.0 * (d + 1) / (ms / n * 1e-3) / 1e12
    print(f"N={N} d={d} R={R} chunk={chunk}: km_step {ms / n:.4f} ms  {tf:.1f} TF  frac {tf / 157.3:.3f}   update+group {msu / max(nu,1):.4f} ms x{nu}", flush=True)
    del x
