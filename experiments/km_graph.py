"""Does a captured graph of the k-means job (63 launches + a few small ones) run faster than the same launches issued one by
one?  configs[1]'s shape, both arithmetics.  usage: python3 experiments/km_graph.py"""
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
asg = torch.empty((R, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()

def job():
    kmeans.kmeans_restarts_dev(ctx, x, N, D, R, 20240, iters, chunk, asg, None, inert)

def timed(f, reps=5):
    f(); torch.cuda.synchronize(); ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    ctx.synchronize(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for arith in (kmeans.ARITH_F32, kmeans.ARITH_BF16X3):
    kmeans.set_arith(ctx, arith)
    direct = timed(job)
    ref = inert.clone()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        ctx.use_torch_stream(True)
        job(); torch.cuda.synchronize()                       # buffers sized, attributes set
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            job()
        ctx.use_torch_stream(False)
    torch.cuda.synchronize()
    inert.zero_()
    graph = timed(g.replay)
    print(["f32", "bf16x3"][arith], "direct %.3f ms, graph replay %.3f ms, same inertia bits: %s" % (direct, graph, bool(torch.equal(ref, inert))))
ctx.close()
