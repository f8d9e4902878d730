"""abw_sig_build on host buffers (PCIe both ways): ABW_PIN_COPIES=0/1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from abawaca_amd import _lib, signature
ctx = _lib.Context(0)
blob, off = bench.make_assembly_blob(20244, 200)
signature.build_signatures(ctx, blob[:10_000_000], off[:1001], want_dims=False)
for rep in range(3):
    t0 = time.perf_counter()
    r = signature.build_signatures(ctx, blob, off, want_dims=False)
    dt = time.perf_counter() - t0
    print(os.environ.get("ABW_PIN_COPIES", "0"), "Gbp/s", round(len(blob) / dt / 1e9, 2), "ms", round(dt * 1e3, 2), "checksum", int(r["milli"].sum()))
