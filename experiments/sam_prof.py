"""abw_sam_parse_dev alone: 2 M generated records (727 MB of SAM text) resident in HBM.  usage: python3 experiments/sam_prof.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from abawaca_amd import _lib, coverage
from _sam_gen import random_sam
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
ctx = _lib.Context(0)
names = ["scaffold_%d" % i for i in range(100_000)]
block = random_sam(20248, 20000, names)
head = int(np.flatnonzero(block == 10)[5]) + 1
text = np.concatenate([block[:head], np.tile(block[head:], 100)])
t_text = torch.from_numpy(text).to(dev)
blob, off = coverage.names_table(names)
cap = 2_000_008
t_s, t_p, t_l = torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int64, device=dev), torch.empty(cap, dtype=torch.int32, device=dev)
cnt = coverage.SamCounts()
pv = lambda t: C.c_void_p(t.data_ptr()); pn = lambda a: C.c_void_p(a.ctypes.data)
def step():
    ctx.check(_lib.lib().abw_sam_parse_dev(ctx.handle, pv(t_text), len(text), pn(blob), pn(off), len(names), 15, cap, pv(t_s), pv(t_p), pv(t_l), None, None, None, None, None, C.byref(cnt)))
step(); ctx.synchronize()
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(3): step()
ctx.synchronize()
n, ms = ctx.prof_read("sam_parse")
print(f"{ms / n:.3f} ms per call, {len(text) / (ms / n * 1e-3) / 1e9:.1f} GB/s, records {cnt.records} kept {cnt.kept}", flush=True)
