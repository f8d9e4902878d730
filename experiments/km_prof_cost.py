"""What do the library's profiling scopes (two HIP events around every km_step / km_update launch group) cost the job they time?
configs[1]'s shape.  usage: python3 experiments/km_prof_cost.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from abawaca_amd import _lib, kmeans

dev = torch.device("cuda", 0)
keys_t, _dp2scaf, _len, _genome, _G = bench.make_points(20241, 200_000, 10, dev, 100)
x = bench.standardise(keys_t)
D, N = x.shape
R, iters = 64, 20
ctx = _lib.Context(0)
chunk = kmeans.auto_chunk(ctx, N, R)
inert = torch.empty(R, dtype=torch.float64, device=dev)
cent = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
asg = torch.empty((R, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()

def job():
    kmeans.kmeans_restarts_dev(ctx, x, N, D, R, 20240, iters, chunk, asg, cent, inert)

def timed(reps=5):
    job(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        job()
    ctx.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for arith in (kmeans.ARITH_F32, kmeans.ARITH_BF16X3):
    kmeans.set_arith(ctx, arith)
    for rnd in range(2):
        ctx.prof_enable(False)
        off = timed()
        ctx.prof_reset(); ctx.prof_enable(True)
        on = timed()
        ctx.prof_enable(False)
        print(["f32", "bf16x3"][arith], "scopes off %.3f ms, on %.3f ms" % (off, on))
ctx.close()
