"""Separator timing by kernel family (the library's own HIP-event brackets): root node + whole recursion.
usage: python3 experiments/sep_prof.py [scaffolds] [samples] [genomes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
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
sep = separator.ClusterSeparatorHip(store)
sep.separate()
ctx.prof_reset(); ctx.prof_enable(True)
t = 0.0
for _ in range(5):
    sep.separate(); t += sep.last_call_s
na, msa = ctx.prof_read("sep_accum"); ne, mse = ctx.prof_read("sep_eval"); nw, msw = ctx.prof_read("sep_wide")
D, N = keys.shape
tab = (msa / na + msw / max(nw, 1)) * 1e-3
print(f"root: {t / 5 * 1e3:.3f} ms per call; accum {msa / na:.4f} ms, wide eval {msw / max(nw, 1):.4f} ms, eval {mse / ne:.4f} ms; algorithmic {(4.0 * D * N + 4 * N + 16 * D * 1001) / tab / 1e12:.3f} TB/s = {(4.0 * D * N + 4 * N + 16 * D * 1001) / tab / 8e12:.3f} of the roof", flush=True)
ctx.prof_reset()
t0 = time.perf_counter()
nodes, s2c, _ = separator.binning(store)
dt = time.perf_counter() - t0
na, msa = ctx.prof_read("sep_accum"); ne, mse = ctx.prof_read("sep_eval")
print(f"recursion: {len(nodes)} nodes, {dt:.3f} s wall, C ABI {sum(r['call_s'] for r in nodes):.3f} s; accum {msa:.1f} ms, eval {mse:.1f} ms total")
# the same recursion in one call (abw_sep_tree)
separator.binning_tree(store, max_nodes=3)
ctx.prof_reset()
best = None
for _ in range(3):
    rows, s2c_t, dt = separator.binning_tree(store)
    best = dt if best is None else min(best, dt)
na, msa = ctx.prof_read("sep_accum"); nw, msw = ctx.prof_read("sep_wide"); ne, mse = ctx.prof_read("sep_eval"); npp, msp = ctx.prof_read("sep_partition")
same = [(r["id"], r["split"], r["dim"], r["key"], r["n1"], r["n2"]) for r in rows] == [(r["id"], r["split"], r["dim"], r["key"], r["n1"], r["n2"]) for r in nodes]
print(f"tree: {len(rows)} nodes in {best * 1e3:.2f} ms (best of 3), identical to node-by-node: {same and bool(np.array_equal(s2c, s2c_t))}; "
      f"per run: narrow accum+eval {msa / 3:.2f} ms, wide {msw / 3:.2f} ms, winner+assign {mse / 3:.2f} ms, partition {msp / 3:.2f} ms over {na // 3} levels")
