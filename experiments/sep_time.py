"""Separator-only run for rocprofv3: the root node of the bench table (configs[1] shape), a few repetitions.
usage: rocprofv3 --kernel-trace --stats ... -- python3 experiments/sep_time.py [scaffolds] [samples]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from abawaca_amd import _lib, separator

n_scaf = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device("cuda", 0)
keys_t, dp2scaf, scaf_len, genome, G = bench.make_points(20241, n_scaf, samples, dev)
ctx = _lib.Context(0)
store = separator.ColumnStore(ctx, keys_t.cpu().numpy(), dp2scaf, scaf_len)
sep = separator.ClusterSeparatorHip(store)
sep.separate()
t = 0.0
for _ in range(5):
    sep.separate()
    t += sep.last_call_s
print("ms per node", t / 5 * 1e3, "split", sep._res.split, "dim", sep._res.dimension)

# where the rest of the call goes: the same node without the host outputs
import ctypes as C
import numpy as np
L = _lib.lib()
out = _lib.Split()
per = (_lib.DimBest * store.D)()
side = np.zeros(store.S, np.uint8); raw = np.zeros(store.N, np.uint8)
def run(per_, side_, raw_, reps=5):
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.check(L.abw_sep_node(ctx.handle, None, store.S, 100, C.byref(out), per_, side_, raw_))
    return (time.perf_counter() - t0) / reps * 1e3
pv = lambda a: C.c_void_p(a.ctypes.data)
print("full        ", run(per, pv(side), pv(raw)))
print("no raw      ", run(per, pv(side), None))
print("no raw/side ", run(per, None, None))
print("no outputs  ", run(None, None, None))
ids = np.arange(store.S, dtype=np.int32)
t0 = time.perf_counter()
for _ in range(5):
    ctx.check(L.abw_sep_node(ctx.handle, pv(ids), store.S, 100, C.byref(out), per, pv(side), pv(raw)))
print("explicit ids", (time.perf_counter() - t0) / 5 * 1e3)
