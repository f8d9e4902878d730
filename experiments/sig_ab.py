"""Signature counting alone on a 1 Gbp synthetic assembly, ms per pass (kernel brackets) + a checksum of the rows: for old / new
library comparisons in one call."""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from abawaca_amd import _lib, signature
dev = torch.device("cuda", 0)
blob, off = bench.make_assembly_blob(20244, 1000)
total = len(blob); pad = (-total) % 16 + 16
t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros(pad, np.uint8)])).to(dev)
t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
cap = total // 2000 + len(off) + 16
t_scaf = torch.empty(cap, dtype=torch.int32, device=dev); t_st = torch.empty(cap, dtype=torch.int64, device=dev); t_en = torch.empty(cap, dtype=torch.int64, device=dev)
t_dims = torch.empty((cap, 180), dtype=torch.float64, device=dev)
torch.cuda.synchronize()
ctx = _lib.Context(0)
n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims); ctx.synchronize()
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(10): signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims)
ctx.synchronize()
k, ms = ctx.prof_read("sig_count")
h = t_dims[:n].view(torch.int64).sum().item()
print(os.environ.get("ABW_LIB_PATH", "")[-10:], n, "segments, %.4f ms per pass" % (ms / k), "rows checksum", h)
