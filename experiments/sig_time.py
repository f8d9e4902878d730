"""Signature-only timing (kernel families via the library's event hooks); ABW_LIB_PATH picks a variant build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from abawaca_amd import _lib, signature
dev = torch.device("cuda", 0)
ctx = _lib.Context(0)
blob, off = bench.make_assembly_blob(20244, 1000)
total = len(blob)
t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-total) % 16 + 16, np.uint8)])).to(dev)
t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
cap = total // 2000 + len(off) + 16
t_scaf = torch.empty(cap, dtype=torch.int32, device=dev); t_st = torch.empty(cap, dtype=torch.int64, device=dev); t_en = torch.empty(cap, dtype=torch.int64, device=dev)
t_dims = torch.empty((cap, 180), dtype=torch.float64, device=dev)
n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims); ctx.synchronize()
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
    signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims)
ctx.synchronize()
nc, msc = ctx.prof_read("sig_count"); ns, mss = ctx.prof_read("sig_segment")
print(os.environ.get("ABW_LIB_PATH", "default"), "count ms", round(msc / nc, 4), "segment ms", round(mss / ns, 4), "checksum", float(t_dims[:n].sum()))
