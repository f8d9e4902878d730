"""Whole recursion on a bench table, for rocprofv3.  usage: ... -- python3 experiments/rec_time.py [scaffolds] [samples]"""
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
t0 = time.perf_counter()
nodes, s2c, d2c = separator.binning(store)
dt = time.perf_counter() - t0
print("recursion s", round(dt, 3), "c_abi_s", round(sum(r["call_s"] for r in nodes), 3), "nodes", len(nodes))
