"""abw_lrn_parse kernels alone on the bench's 200 000 x 189 table (234 MB of text): for old / new library comparisons."""
import os, sys, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abawaca_amd import lrn
from abawaca_amd._lib import Context
rng = np.random.default_rng(7)
N, D = 200_000, 189
m = rng.integers(0, 1001, size=(N, D)).astype(np.int32)
m[:, -10:] = rng.integers(0, 400_000, size=(N, 10))
ctx = Context(0)
rows = lrn.lrn_emit(ctx, m, 1)
head = (f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\tc{j}" for j in range(D)) + "\n").encode()
text = np.concatenate([np.frombuffer(head, np.uint8), rows])
names, keys, _ = lrn.lrn_parse(ctx, text)
assert np.array_equal(keys.T, m)
ctx.prof_reset(); ctx.prof_enable(True)
for _ in range(5):
    lrn.lrn_parse(ctx, text)
_, ms = ctx.prof_read("lrn_parse"); ctx.prof_enable(False)
print(f"{os.environ.get('ABW_LIB_PATH', 'in-tree')}: {ms / 5:.3f} ms kernels for {len(text) / 1e6:.1f} MB, {len(text) / (ms / 5) / 1e6:.0f} GB/s of text", flush=True)
