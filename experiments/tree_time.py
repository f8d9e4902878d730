"""abw_sep_tree on a bench table, for rocprofv3.  usage: ... -- python3 experiments/tree_time.py [scaffolds] [samples] [genomes]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from abawaca_amd import _lib, separator
n_scaf = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 10
genomes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
dev = torch.device("cuda", 0)
keys_t, dp2scaf, scaf_len, genome, G = bench.make_points(20241, n_scaf, samples, dev, genomes)
keys = keys_t.cpu().numpy(); del keys_t
scg_scaf, scg_id = bench.make_scgs(20241, genome, G)
so, si = separator.scg_csr(scg_scaf, scg_id, n_scaf)
ctx = _lib.Context(0)
store = separator.ColumnStore(ctx, keys, dp2scaf, scaf_len, so, si, 30)
separator.binning_tree(store, max_nodes=3)
best = None
for _ in range(3):
    rows, s2c, dt = separator.binning_tree(store)
    best = dt if best is None else min(best, dt)
print("tree", len(rows), "nodes", round(best * 1e3, 2), "ms")
