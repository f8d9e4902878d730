import sys, torch
sys.path.insert(0, "/root/repo")
from abawaca_amd import _lib, kmeans
N, d, R, iters = 799859, 189, 64, 20
x = torch.randn(d, N, device="cuda")
ctx = _lib.Context(0)
chunk = kmeans.auto_chunk(ctx, N, R)
cent = torch.empty(R * 2 * d, device="cuda"); inert = torch.empty(R, dtype=torch.float64, device="cuda")
asg = torch.empty((R, N), dtype=torch.uint8, device="cuda")
for rep in range(2):
    ctx.prof_reset(); ctx.prof_enable(True)
    kmeans.kmeans_restarts_dev(ctx, x, N, d, R, 1234, iters, chunk, asg, cent, inert)
    ctx.synchronize()
    print("chunk", chunk, {k: ctx.prof_read(k) for k in ("km_step", "km_update", "km_moments")})
    ctx.prof_enable(False)
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
kmeans.kmeans_restarts_dev(ctx, x, N, d, R, 1234, iters, chunk, asg, cent, inert)
ctx.synchronize(); e.record(); torch.cuda.synchronize()
print("total ms (torch stream events, approx)", s.elapsed_time(e))
