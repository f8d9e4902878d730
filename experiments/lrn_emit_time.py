"""abw_lrn_emit (the build writer): kernel time for 200 000 rows x 189 columns -- the bench's figure alone."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abawaca_amd import lrn
from abawaca_amd._lib import Context
rng = np.random.default_rng(5)
m = rng.integers(0, 400_000, size=(200_000, 189)).astype(np.int32)
m[::97] *= -1
ctx = Context(0)
t = lrn.lrn_emit(ctx, m, 1)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    lrn.lrn_emit(ctx, m, 1)
_, ms = ctx.prof_read("lrn_emit"); ctx.prof_enable(False)
print(f"{os.environ.get('ABW_LIB_PATH', 'in-tree')}: {ms / 5:.3f} ms kernels, {len(t) / (ms / 5) / 1e6:.0f} GB/s of text, sha1 {hashlib.sha1(t.tobytes()).hexdigest()[:12]}", flush=True)
