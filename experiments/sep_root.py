"""Root node of a bench table through abw_sep_node (no per-dimension output, as the drop-in adapter calls it), brackets printed.
usage: [rocprofv3 --kernel-trace --stats -d DIR --] python3 experiments/sep_root.py [scaffolds] [samples] [genomes] [frac]
frac < 1: also a node holding every scaffold of the first frac x genomes (a big node below the root)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from abawaca_amd import _lib, separator
n_scaf = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 10
genomes = int(sys.argv[3]) if len(sys.argv) > 3 else 0
frac = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
dev = torch.device("cuda", 0)
keys_t, dp2scaf, scaf_len, genome, G = bench.make_points(20241, n_scaf, samples, dev, genomes)
keys = keys_t.cpu().numpy(); del keys_t
scg_scaf, scg_id = bench.make_scgs(20241, genome, G)
so, si = separator.scg_csr(scg_scaf, scg_id, n_scaf)
ctx = _lib.Context(0)
n_scg = 0 if os.environ.get("NO_SCG") else 30
store = separator.ColumnStore(ctx, keys, dp2scaf, scaf_len, so if n_scg else None, si if n_scg else None, n_scg)
D, N = keys.shape
L = _lib.lib()
out = _lib.Split()
side = np.zeros(store.S, np.uint8)
pv = lambda a: C.c_void_p(a.ctypes.data)
def run(ids, label, n_dps):
    m = store.S if ids is None else len(ids)
    p = None if ids is None else pv(ids)
    for per in ((None,) if os.environ.get("NO_PER_DIM_ONLY") else (None, (_lib.DimBest * D)())):
        ctx.check(L.abw_sep_node(ctx.handle, p, m, 100, C.byref(out), per, pv(side), None))
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(5):
            ctx.check(L.abw_sep_node(ctx.handle, p, m, 100, C.byref(out), per, pv(side), None))
        dt = (time.perf_counter() - t0) / 5
        na, msa = ctx.prof_read("sep_accum"); ne, mse = ctx.prof_read("sep_eval"); nw, msw = ctx.prof_read("sep_wide")
        ctx.prof_enable(False)
        tab = (msa / na + msw / max(nw, 1)) * 1e-3
        alg = 4.0 * D * n_dps + 4 * n_dps + 16 * D * 1001
        print(f"{label} per_dim={'yes' if per else 'no'}: call {dt * 1e3:.3f} ms; accum {msa / na:.4f}, wide eval {msw / max(nw, 1):.4f}, eval {mse / ne:.4f} ms; "
              f"{alg / tab / 1e12:.3f} TB/s = {alg / tab / 8e12:.3f} of the roof; split {out.split} dim {out.dimension} key {out.key}", flush=True)
run(None, "root", N)
if frac < 1.0:
    ids = np.flatnonzero(genome < int(G * frac)).astype(np.int32)
    run(ids, f"node of {len(ids)} scaffolds", int(store.scaf_ndps[ids].sum()))
