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
