"""Cluster writer (abw_lrn_emit_rows_dev): 100 clusters of 2 000 rows x 189 columns one call each, and the same rows in ONE
call -- the bench's figures, for old/new library comparisons in one gpurun call (ABW_LIB_PATH picks the library)."""
import os, sys, time, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abawaca_amd import lrn
from abawaca_amd._lib import Context

rng = np.random.default_rng(5)
N, D = 200_000, 189
vals = (rng.integers(0, 400_000, size=(D, N)) / 1000.0).astype(np.float64)
vals[:, ::97] *= -1.0
ctx = Context(0)
t_vals = torch.from_numpy(vals).cuda()
perm = torch.from_numpy(rng.permutation(N).astype(np.int64) + 1).cuda()
n_cl, per = 100, N // 100
t_out = torch.empty(per * (D * 14 + 12), dtype=torch.uint8, device="cuda")
def clusters():
    tot = 0
    for c in range(n_cl):
        tot += lrn.lrn_emit_rows_dev(ctx, t_vals, perm[c * per:(c + 1) * per], t_out)
    return tot
clusters()
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter(); cbytes = clusters(); wall = time.perf_counter() - t0
_, ms = ctx.prof_read("lrn_emit_rows"); ctx.prof_enable(False)
t_all = torch.empty(cbytes, dtype=torch.uint8, device="cuda")
t_off = torch.empty(N + 1, dtype=torch.int64, device="cuda")
lrn.lrn_emit_rows_dev(ctx, t_vals, perm, t_all, t_off)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    lrn.lrn_emit_rows_dev(ctx, t_vals, perm, t_all, t_off)
_, mso = ctx.prof_read("lrn_emit_rows"); ctx.prof_enable(False)
# the same from the resident (transposed) copy
shape = lrn.lrn_table_load_dev(ctx, t_vals)
def clusters_r():
    tot = 0
    for c in range(n_cl):
        tot += lrn.lrn_emit_rows_dev(ctx, None, perm[c * per:(c + 1) * per], t_out, resident=shape)
    return tot
clusters_r()
ctx.prof_reset(); ctx.prof_enable(True)
t0 = time.perf_counter(); rbytes = clusters_r(); wall_r = time.perf_counter() - t0
_, msr = ctx.prof_read("lrn_emit_rows"); ctx.prof_enable(False)
t_all2 = torch.empty(cbytes, dtype=torch.uint8, device="cuda")
lrn.lrn_emit_rows_dev(ctx, None, perm, t_all2, t_off, resident=shape)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    lrn.lrn_emit_rows_dev(ctx, None, perm, t_all2, t_off, resident=shape)
_, msro = ctx.prof_read("lrn_emit_rows"); ctx.prof_enable(False)
assert rbytes == cbytes and torch.equal(t_all, t_all2)
print(f"resident copy: per cluster {msr / n_cl * 1e3:.1f} us kernels, {wall_r / n_cl * 1e6:.1f} us wall; one call {msro / 5:.3f} ms = {cbytes / (msro / 5) / 1e6:.0f} GB/s", flush=True)
h = hashlib.sha1(t_all.cpu().numpy().tobytes()).hexdigest()[:12]
ho = hashlib.sha1(t_off.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"{os.environ.get('ABW_LIB_PATH', 'in-tree')}: per cluster {ms / n_cl * 1e3:.1f} us kernels, {wall / n_cl * 1e6:.1f} us wall; "
      f"one call {mso / 5:.3f} ms = {cbytes / (mso / 5) / 1e6:.0f} GB/s; text {cbytes} sha1 {h} offsets {ho}", flush=True)
