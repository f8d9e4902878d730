"""K-means extension against an INDEPENDENT plain float64 Lloyd (tests/lloyd_textbook.py: two distances, two means) on
configs[1]-shaped data, all 20 iterations: the documented tolerance of DESIGN.md §3.5.  The library runs one fp32
discriminant per point on the MFMA and rounds its centroids to fp32 every iteration; the textbook run keeps doubles.
Assignments may therefore differ at near-ties and the trajectories drift apart slightly; this test bounds both:

  * after EVERY iteration (0 = assignment by the seed points ... 20) at most FLIP_TOL of the points sit in the other cluster
  * the final inertia agrees to INERTIA_TOL (relative)

(The bit-exact check is against oracle/abw_oracle.c, which restates the library's own fp32 arithmetic --
tests/test_gpu_kmeans.py; the reference itself has no k-means: parity unpinned.)"""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import lloyd_textbook as lt  # noqa: E402
from abawaca_amd import _lib, kmeans  # noqa: E402

pytestmark = pytest.mark.gpu

FLIP_TOL = 1e-5        # share of points in the other cluster than the float64 run, any iteration, any restart (measured r2: 0)
INERTIA_TOL = 1e-6     # relative difference of the final inertia (measured r2: 4.1e-8)


def _against_float64_lloyd(samples, R, check, arith, outname, scaffolds=200_000):
    import torch
    dev = torch.device("cuda", 0)
    keys_t, _dp2scaf, _len, _genome, _G = bench.make_points(20241, scaffolds, samples, dev)
    x_t = bench.standardise(keys_t)                               # fp32 [179 + samples, N], what bench.py clusters
    torch.cuda.synchronize()          # the library runs on its own stream: x_t must be complete before it reads it
    D, N = x_t.shape
    x64 = x_t.to(torch.float64)
    iters, seed = 20, 20240
    ctx = _lib.Context(0)
    try:
        chunk = kmeans.auto_chunk(ctx, N, R)
        kmeans.set_arith(ctx, arith)
        # the library's assignment after `it` iterations, for every it (a run of `it` iterations ends with an assign pass)
        lib_assign, lib_inertia, lib_cent = [], None, None
        for it in range(iters + 1):
            a = torch.empty((R, N), dtype=torch.uint8, device=dev)
            i = torch.empty(R, dtype=torch.float64, device=dev)
            c = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
            kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R, seed, it, chunk, a, c, i)
            ctx.synchronize()
            lib_assign.append(a[check].clone())
            lib_inertia, lib_cent = i, c
        if arith != kmeans.ARITH_F32:
            # the split arithmetic really ran: its centroids are close to the exact path's, and not the same bits
            kmeans.set_arith(ctx, kmeans.ARITH_F32)
            c32 = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
            kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R, seed, iters, chunk, None, c32, None)
            ctx.synchronize()
            assert not torch.equal(c32, lib_cent)
        worst_flip, worst_inertia, flips = 0.0, 0.0, []
        for k, r in enumerate(check):
            row = []
            for it, mem, _c0, _c1, inert in lt.lloyd(torch, x64, seed, r, iters):
                frac = float((mem != lib_assign[it][k].to(torch.bool)).double().mean())
                row.append(frac)
            flips.append(row)
            worst_flip = max(worst_flip, max(row))
            worst_inertia = max(worst_inertia, abs(inert - float(lib_inertia[r])) / inert)
            assert 0 < int(mem.sum()) < N
        out = os.path.join(ROOT, "gpurun_out")
        if os.path.isdir(out):
            json.dump(dict(N=N, D=D, R=R, restarts_checked=list(check), arith=["f32", "bf16x3"][arith], iters=iters, chunk=chunk,
                           worst_flip_fraction=worst_flip, worst_inertia_rel=worst_inertia,
                           flip_fraction_per_restart_and_iteration=flips), open(os.path.join(out, outname), "w"))
        assert worst_flip <= FLIP_TOL, flips
        assert worst_inertia <= INERTIA_TOL
    finally:
        ctx.close()


def test_twenty_iterations_against_float64_lloyd():
    _against_float64_lloyd(10, 4, [0, 1, 2, 3], kmeans.ARITH_F32, "lloyd_tolerance.json")


def test_split_bf16_arithmetic_keeps_the_same_tolerance_at_configs1():
    """abw_km_set_arith(ABW_KM_ARITH_BF16X3), the shape of configs[1] (64 restarts = two restart tiles per workgroup, 189
    dims): restarts of both tiles, first, middle and last lanes."""
    _against_float64_lloyd(10, 64, [0, 1, 31, 32, 62, 63], kmeans.ARITH_BF16X3, "lloyd_tolerance_bf16x3_c1.json")


def test_split_bf16_arithmetic_keeps_the_same_tolerance_at_199_dims():
    """The other kernel shape (four restart tiles per workgroup, 199 dims as configs[2]): 128 restarts, one checked per tile."""
    _against_float64_lloyd(20, 128, [0, 33, 70, 127], kmeans.ARITH_BF16X3, "lloyd_tolerance_bf16x3_199.json")


@pytest.mark.parametrize("arith", [kmeans.ARITH_F32, kmeans.ARITH_BF16X3])
def test_configs2_size_against_float64_lloyd(arith):
    """configs[2] at its stated size -- 1M scaffolds -> 4.0 M points x 199 dims, ONE job of 256 restarts x 20 iterations -- in
    both arithmetics: three restarts (first, middle, last restart tile) followed through every iteration by the float64
    textbook run, same bounds."""
    _against_float64_lloyd(20, 256, [0, 100, 255], arith, "lloyd_tolerance_c2_%s.json" % ["f32", "bf16x3"][arith], scaffolds=1_000_000)
