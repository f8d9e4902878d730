"""GPU checks of the k-means restart extension against its CPU restatement (oracle/abw_or_kmeans)
and of the device TinyMT32 against the reference's known answers.  The reference has no k-means:
parity here is "unpinned"  except for the RNG."""
import json
import os

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, kmeans

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def test_device_tinymt_matches_reference_kat(ctx, golden_dir):
    kat = json.load(open(os.path.join(golden_dir, "tinymt32_kat.json")))
    for seed, outs in kat["outputs"].items():
        assert kmeans.tinymt_selftest(ctx, int(seed), len(outs)) == outs


def _blobs(seed, N, d, sep=4.0):
    rng = np.random.default_rng(seed)
    lab = rng.integers(0, 2, size=N)
    mu = rng.normal(0, 1, size=(2, d)) * sep
    x = (mu[lab] + rng.normal(0, 1, size=(N, d))).astype(np.float32)
    return np.ascontiguousarray(x.T), lab


@pytest.mark.parametrize("N,d,R,iters,chunk", [(1000, 7, 5, 4, 64), (4097, 33, 16, 6, 256), (6000, 189, 64, 3, 512),
                                                  (333, 31, 3, 2, 32), (2500, 64, 20, 3, 128),
                                                  (3000, 150, 8, 3, 256), (2100, 199, 32, 2, 160)])   # one LDS buffer per point lane
def test_kmeans_bit_exact_vs_oracle(ctx, N, d, R, iters, chunk):
    x, _ = _blobs(N + d, N, d)
    a, c, inert = kmeans.kmeans_restarts(ctx, x, R, base_seed=20240, iters=iters, chunk=chunk)
    oa, oc, oi = orc.kmeans(x, R, 20240, iters, chunk)
    assert np.array_equal(c.view(np.uint32), oc.view(np.uint32))        # centroids: identical fp32 bits
    assert np.array_equal(a, oa)                                         # assignments identical
    assert np.array_equal(inert, oi)                                     # fp64 inertia identical


@pytest.mark.parametrize("N,d,R,iters,chunk", [(6000, 189, 64, 3, 512),      # two restart tiles, two point lanes (configs[1]'s kernel)
                                                  (2101, 130, 40, 3, 96),       # ragged N, padded restarts, last chunk short
                                                  (5003, 199, 100, 3, 160),     # four restart tiles, the last one partly padding
                                                  (4000, 150, 96, 2, 128),      # three restart tiles: one wave of the workgroup idle
                                                  (3000, 189, 256, 2, 256)])    # two workgroup rows of restarts
def test_split_bf16_arithmetic_agrees_with_the_exact_path(ctx, N, d, R, iters, chunk):
    """abw_km_set_arith(ABW_KM_ARITH_BF16X3) on every kernel shape it has, ragged edges included: the Lloyd passes run on three
    bf16 planes, so the bits differ from the exact path's -- centroids to 1e-5 of the data's scale, inertia to 1e-6, and on
    separated blobs the same assignments but for near ties."""
    x, _ = _blobs(N + d, N, d)
    a0, c0, i0 = kmeans.kmeans_restarts(ctx, x, R, base_seed=20240, iters=iters, chunk=chunk)
    kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
    try:
        a1, c1, i1 = kmeans.kmeans_restarts(ctx, x, R, base_seed=20240, iters=iters, chunk=chunk)
    finally:
        kmeans.set_arith(ctx, kmeans.ARITH_F32)
    assert not np.array_equal(c0.view(np.uint32), c1.view(np.uint32))   # the other arithmetic really ran
    scale = float(np.abs(x).max())
    assert np.abs(c1.astype(np.float64) - c0.astype(np.float64)).max() <= 1e-5 * scale
    assert (np.abs(i1 - i0) / i0).max() <= 1e-6
    assert (a0 != a1).sum(axis=1).max() <= max(1, N // 1000)
    with pytest.raises(_lib.AbwError):
        kmeans.set_arith(ctx, 7)


@pytest.mark.parametrize("N,d,R", [(3000, 189, 16), (3000, 60, 64), (3000, 199, 64)])
def test_split_bf16_arithmetic_runs_the_exact_kernel_where_it_has_none(ctx, N, d, R):
    """One restart tile, fewer than 128 columns, or 64 restarts with more than 191 dims: no split kernel -- the exact one runs
    and the bits are the exact path's."""
    x, _ = _blobs(N + d, N, d)
    a0, c0, i0 = kmeans.kmeans_restarts(ctx, x, R, base_seed=7, iters=3, chunk=128)
    kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
    try:
        a1, c1, i1 = kmeans.kmeans_restarts(ctx, x, R, base_seed=7, iters=3, chunk=128)
    finally:
        kmeans.set_arith(ctx, kmeans.ARITH_F32)
    assert np.array_equal(c0.view(np.uint32), c1.view(np.uint32)) and np.array_equal(a0, a1) and np.array_equal(i0, i1)


def test_kmeans_recovers_planted_clusters(ctx):
    x, lab = _blobs(7, 20000, 50, sep=3.0)
    a, c, inert = kmeans.kmeans_restarts(ctx, x, 8, base_seed=1, iters=10)
    best = int(np.argmin(inert))
    agree = (a[best] == lab).mean()
    assert max(agree, 1 - agree) > 0.99
    # inertia equals the direct sum of squared distances (fp64) within rounding
    pts = x.T.astype(np.float64)
    direct = ((pts - c[best][a[best]].astype(np.float64)) ** 2).sum()
    assert abs(direct - inert[best]) / direct < 1e-6


def test_argument_and_state_errors_are_reported_not_faulted(ctx):
    import ctypes as C
    L = _lib.lib()
    x = np.zeros((4, 16), dtype=np.float32)
    a = np.zeros((1, 16), dtype=np.uint8)
    # fewer than 32 points, too many dimensions, bad chunk: error codes, message available, no launch
    rc = L.abw_km_run(ctx.handle, C.c_void_p(x.ctypes.data), 16, 4, 1, 1, 1, 0, C.c_void_p(a.ctypes.data), None, None)
    assert rc == _lib.ABW_ERR_ARG and b"32" in L.abw_last_error(ctx.handle)
    big = np.zeros((224, 64), dtype=np.float32)
    rc = L.abw_km_run(ctx.handle, C.c_void_p(big.ctypes.data), 64, 224, 1, 1, 1, 0, None, None, None)
    assert rc == _lib.ABW_ERR_ARG and b"223" in L.abw_last_error(ctx.handle)
    ok = np.zeros((4, 64), dtype=np.float32)
    rc = L.abw_km_run(ctx.handle, C.c_void_p(ok.ctypes.data), 64, 4, 1, 1, 1, 48, None, None, None)
    assert rc == _lib.ABW_ERR_ARG and b"multiple of 32" in L.abw_last_error(ctx.handle)
    # phase calls without a run in progress
    assert L.abw_km_shard_step(ctx.handle, None) == _lib.ABW_ERR_STATE
    assert L.abw_km_shard_update(ctx.handle, 0) == _lib.ABW_ERR_STATE
    assert L.abw_km_shard_finish(ctx.handle, None, None) == _lib.ABW_ERR_STATE
    # identical points: every restart collapses to one cluster, nothing divides by zero
    same = np.ones((3, 96), dtype=np.float32)
    asg, cen, ine = kmeans.kmeans_restarts(ctx, same, 2, base_seed=3, iters=2, chunk=32)
    assert np.all(asg == 0) and np.all(ine == 0.0) and np.all(cen == 1.0)


def test_a_captured_graph_of_the_job_replays_to_the_same_bits():
    """abw_km_run_dev allocates nothing once its buffers are sized and only launches on the context's stream, so a caller may
    capture the whole job -- the 21 passes' launches -- into a HIP graph and replay it (experiments/km_graph.py: 1 - 4 % faster than
    issuing them; the gaps between dependent kernels are not the launches')."""
    import torch
    dev = torch.device("cuda", 0)
    N, d, R, iters = 40_000, 150, 64, 4
    x = torch.randn((d, N), device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    x[:, : N // 2] += 1.5
    inert = torch.empty(R, dtype=torch.float64, device=dev)
    asg = torch.empty((R, N), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    c = _lib.Context(0)
    try:
        def job():
            kmeans.kmeans_restarts_dev(c, x, N, d, R, 11, iters, 512, asg, None, inert)
        job(); c.synchronize()
        want_i, want_a = inert.clone(), asg.clone()
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            c.use_torch_stream(True)
            job(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                job()
            c.use_torch_stream(False)
        torch.cuda.synchronize()
        for _ in range(2):
            inert.zero_(); asg.zero_()
            g.replay(); torch.cuda.synchronize()
            assert torch.equal(inert, want_i) and torch.equal(asg, want_a)
    finally:
        c.close()


def test_the_64_bit_pointer_staging_path_gives_the_same_bits():
    """Tables whose row offsets do not fit 32 bits ((d - 1) N >= 2^30) take the step kernel's other staging path (64-bit pointers
    instead of a buffer descriptor + 32-bit offsets).  ABW_KM_PTR64=1 forces it on a small table, in a process of its own (the
    switch is read once): bit-identical to the CPU restatement there too."""
    import subprocess
    import sys
    code = (
        "import sys, numpy as np\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import _oracle as orc\n"
        "from abawaca_amd import _lib, kmeans\n"
        "rng = np.random.default_rng(3)\n"
        "for N, d, R, chunk in ((6000, 189, 64, 512), (2100, 199, 32, 160), (4097, 33, 16, 256)):\n"
        "    x = np.ascontiguousarray((rng.normal(0, 1, (N, d)) + 3.0 * rng.integers(0, 2, (N, 1))).astype(np.float32).T)\n"
        "    c = _lib.Context(0)\n"
        "    a, ce, i = kmeans.kmeans_restarts(c, x, R, base_seed=20240, iters=3, chunk=chunk)\n"
        "    oa, oc, oi = orc.kmeans(x, R, 20240, 3, chunk)\n"
        "    assert np.array_equal(ce.view(np.uint32), oc.view(np.uint32)) and np.array_equal(a, oa) and np.array_equal(i, oi)\n"
        "    c.close()\n"
        "print('same bits')\n"
    ) % (os.path.dirname(os.path.abspath(__file__)), os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    env = dict(os.environ, ABW_KM_PTR64="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "same bits" in p.stdout, p.stderr[-2000:]
