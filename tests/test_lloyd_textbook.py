"""CPU: the textbook Lloyd of tests/lloyd_textbook.py (numpy, float64) on small inputs -- its TinyMT32 against the
reference's known answers, and the library's semantics (as restated by the CPU oracle) against it: same seed points, and
on well-separated data identical assignments."""
import json
import os

import numpy as np

import _oracle as orc
import lloyd_textbook as lt


def test_pure_python_tinymt_matches_reference_kat(golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "tinymt32_kat.json")))
    for seed, outs in kat["outputs"].items():
        assert lt.tinymt32_stream(int(seed), len(outs)) == outs


def test_oracle_kmeans_follows_textbook_lloyd_on_small_data():
    rng = np.random.default_rng(3)
    d, N = 12, 6000
    x = rng.normal(size=(d, N)).astype(np.float32)
    x[:, : N // 3] += 2.5                                        # two blobs
    R, iters, seed = 6, 8, 90
    a, c, inertia = orc.kmeans(x, R, seed, iters, 256)
    x64 = x.astype(np.float64)
    for r in range(R):
        for it, mem, c0, c1, inert in lt.lloyd(np, x64, seed, r, iters):
            pass
        assert float((mem != a[r].astype(bool)).mean()) <= 2e-3  # near-tie flips only
        assert abs(inert - inertia[r]) / inert < 1e-5
        assert np.allclose(c[r, 0], c0, atol=2e-3) and np.allclose(c[r, 1], c1, atol=2e-3)
