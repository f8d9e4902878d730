"""Per-phase cycle accounting of km_step_kernel (wave 0 of workgroup 7), library built with -DABW_KM_TIMING."""
import ctypes, os, sys, numpy as np
os.environ.setdefault("ABW_LIB_PATH", os.path.join(os.path.dirname(__file__), "lib_TIMING.so"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from abawaca_amd import _lib, kmeans
N, d, R, iters = 799859, 189, 64, 5
g = torch.Generator(device="cuda").manual_seed(1)
x = torch.randn(d, N, device="cuda", generator=g)
ctx = _lib.Context()
lib = _lib.lib()
out = (ctypes.c_ulonglong * 10)()
lib.abw_dbg_km_timing(out)
cent = torch.empty(R * 2 * d, device="cuda"); inert = torch.empty(R, dtype=torch.float64, device="cuda")
kmeans.kmeans_restarts_dev(ctx, x, N, d, R, 1234, iters, 1568, None, cent, inert)
torch.cuda.synchronize()
lib.abw_dbg_km_timing(out)
v = np.array(list(out), dtype=np.float64)
print("core clock during km_step: %.0f MHz" % (v[:8].sum() / v[8] * 100.0))
v = v[:8]
names = ["pre-chunk", "chunk-setup", "stage_load", "mfma1-issue", "membership(+drain)", "mfma2-issue", "store+barrier", "slab-store"]
launches = iters + 1
ntile = 50 * launches
print("total cycles/launch %.0f" % (v.sum() / launches))
for n, c in zip(names, v):
    print("%-20s %10.0f /launch  %8.1f /tile" % (n, c / launches, c / ntile))
