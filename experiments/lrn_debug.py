import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from abawaca_amd import _lib, lrn
import test_gpu_lrn as T
ctx = _lib.Context(0)
text, names, milli = T._table_text(np.random.default_rng(5000), 5000, 189)
got_names, keys, dims = lrn.lrn_parse(ctx, text)
bad = np.flatnonzero(got_names != names)
print("name mismatches", len(bad), bad[:10], got_names[bad[:10]], names[bad[:10]])
kb = np.argwhere(keys != milli.T)
print("key mismatches", len(kb), kb[:10])
for d, r in kb[:10]: print(d, r, keys[d, r], milli.T[d, r])
# where in the text do the bad rows start?
t = text.tobytes()
starts = [0]
for i, c in enumerate(t):
    if c == 10: starts.append(i + 1)
for r in bad[:10]:
    off = starts[r + 4]
    print("row", r, "offset", off, "block", off // 1008, "in-block", off % 1008, t[off:off + 12])
