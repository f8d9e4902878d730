"""configs[2]-shaped job (d = 199, 256 restarts, 1 M points), wall clock per job: for old / new library comparisons in one call."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
from abawaca_amd import _lib, kmeans
dev = torch.device("cuda", 0)
keys_t, *_ = bench.make_points(20242, 250_000, 20, dev, 200)
x = bench.standardise(keys_t); D, N = x.shape; R, iters = 256, 20
ctx = _lib.Context(0); chunk = kmeans.auto_chunk(ctx, N, R)
if os.environ.get("KM_SPLIT"): kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
inert = torch.empty(R, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
def job(): kmeans.kmeans_restarts_dev(ctx, x, N, D, R, 20240, iters, chunk, None, None, inert)
for _ in range(3): job()
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5): job()
ctx.synchronize()
print(os.environ.get("ABW_LIB_PATH", "")[-10:], D, N, "%.3f ms per job" % ((time.perf_counter() - t0) / 5 * 1e3))
