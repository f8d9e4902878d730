import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from abawaca_amd import _lib, kmeans
dev = torch.device("cuda", 0)
keys_t, *_ = bench.make_points(20241, 200_000, 10, dev, 100)
x = bench.standardise(keys_t); D, N = x.shape; R, iters = 64, 20
ctx = _lib.Context(0); chunk = kmeans.auto_chunk(ctx, N, R)
inert = torch.empty(R, dtype=torch.float64, device=dev); asg = torch.empty((R, N), dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
if os.environ.get("KM_SPLIT"): kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
def job(): kmeans.kmeans_restarts_dev(ctx, x, N, D, R, 20240, iters, chunk, asg, None, inert)
for _ in range(10): job()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(20): job()
ctx.synchronize()
print(os.environ.get("ABW_LIB_PATH", "")[-10:], "%.3f ms per job" % ((time.perf_counter() - t0) / 20 * 1e3))
