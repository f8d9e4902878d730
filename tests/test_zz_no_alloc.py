"""abw_sig_segment_static — the entry point that allocates nothing (its working arrays live in a file-scope device array
of the library) — against abw_sig_segment_dev, through the C ABI: identical segment tables on two calls in a row, the
second with a smaller input (the array keeps the first call's contents), plus the count-only / capacity / limit rules."""
import ctypes as C

import numpy as np
import pytest

from abawaca_amd import _lib, signature, synth

pytestmark = pytest.mark.gpu


def _assembly(seed, n_scaf, lower_n=False):
    rng = np.random.default_rng(seed)
    seqs = []
    for i in range(n_scaf):
        L = int(rng.integers(1, 12000))
        s = synth.ACGT[rng.integers(0, 4, size=L)].copy()
        for _k in range(int(rng.integers(0, 3))):
            a = int(rng.integers(0, L))
            s[a:min(L, a + int(rng.integers(1, 1500)))] = ord("N")
        if i % 7 == 0:
            s[:min(L, 40)] |= 0x20                                 # lower-case acgt
            s[s == ord("n")] = ord("N")
        seqs.append(s)
    seqs.append(np.full(9, ord("N"), dtype=np.uint8))              # all-N scaffold: a segment per base
    if lower_n:
        seqs[3][0] = ord("n")
    blob = np.concatenate(seqs)
    off = np.zeros(len(seqs) + 1, dtype=np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return blob, off


def _both(ctx, blob, off, window=2000, cap=None):
    import torch
    dev = torch.device("cuda", 0)
    total = len(blob)
    t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-total) % 16 + 16, np.uint8)])).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    cap = total // window + len(off) + 32 if cap is None else cap
    ref = [torch.full((cap,), -1, dtype=dt, device=dev) for dt in (torch.int32, torch.int64, torch.int64)]
    n = signature.segment_dev(ctx, t_ascii, total, t_off, window, *ref)
    ctx.synchronize()
    got = [torch.full((cap,), -1, dtype=dt, device=dev) for dt in (torch.int32, torch.int64, torch.int64)]
    res = torch.full((2,), -1, dtype=torch.int64, device=dev)
    signature.segment_static(ctx, t_ascii, total, t_off, window, *got, res)
    ctx.synchronize()
    return n, [t.cpu().numpy() for t in ref], [t.cpu().numpy() for t in got], res.cpu().numpy()


def test_static_entry_matches_segment_dev_twice():
    ctx = _lib.Context(0)
    try:
        for seed, n_scaf in ((1, 3000), (2, 170)):                 # the second call is the smaller one
            blob, off = _assembly(seed, n_scaf)
            n, ref, got, res = _both(ctx, blob, off)
            assert n > n_scaf and res[0] == n and res[1] == 0
            for a, b in zip(ref, got):
                assert np.array_equal(a, b)                        # bit-exact, the untouched tail (-1) included
    finally:
        ctx.close()


def test_static_entry_allocates_nothing_and_does_not_wait():
    """What the entry point is for: the device's free memory (hipMemGetInfo, which sees the library's allocations and
    torch's alike) is the same before and after static calls on inputs and outputs that already exist, also for a larger
    input than any before (the context's grow-only buffers would grow), and the call returns without a synchronisation:
    nothing comes back to the host but the status."""
    import torch
    dev = torch.device("cuda", 0)
    ctx = _lib.Context(0)
    try:
        cases = []
        for seed, n_scaf in ((11, 200), (12, 4000), (13, 900)):    # small (warm-up), larger than anything before, smaller
            blob, off = _assembly(seed, n_scaf)
            total = len(blob)
            t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-total) % 16 + 16, np.uint8)])).to(dev)
            t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
            cap = total // 2000 + len(off) + 32
            outs = got_like(dev, cap)
            res = torch.full((2,), -1, dtype=torch.int64, device=dev)
            cases.append((t_ascii, total, t_off, outs, res, blob, off))
        signature.segment_static(ctx, cases[0][0], cases[0][1], cases[0][2], 2000, *cases[0][3], cases[0][4])   # code object loaded
        ctx.synchronize()
        torch.cuda.synchronize()
        free0 = torch.cuda.mem_get_info(0)[0]
        for t_ascii, total, t_off, outs, res, _b, _o in cases[1:]:
            signature.segment_static(ctx, t_ascii, total, t_off, 2000, *outs, res)
        free1 = torch.cuda.mem_get_info(0)[0]
        ctx.synchronize()
        assert free1 == free0 and torch.cuda.mem_get_info(0)[0] == free0
        for t_ascii, total, t_off, outs, res, blob, off in cases[1:]:       # and the results are the other entry point's
            ref = got_like(dev, len(outs[0]))
            torch.cuda.synchronize()                            # (torch fills on its own stream)
            n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, *ref)
            ctx.synchronize()
            assert int(res[0]) == n and int(res[1]) == 0
            for a, b in zip(ref, outs):
                assert torch.equal(a, b)
    finally:
        ctx.close()


def test_static_entry_rules():
    import torch
    ctx = _lib.Context(0)
    try:
        blob, off = _assembly(3, 50)
        # a capacity below the count: the count still comes back, the first `cap` segments are written
        n, ref, got, res = _both(ctx, blob, off)
        cap = n // 2
        dev = torch.device("cuda", 0)
        total = len(blob)
        t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-total) % 16 + 16, np.uint8)])).to(dev)
        t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        small = [torch.full((cap,), -1, dtype=dt, device=dev) for dt in (torch.int32, torch.int64, torch.int64)]
        r2 = torch.zeros(2, dtype=torch.int64, device=dev)
        signature.segment_static(ctx, t_ascii, total, t_off, 2000, *small, r2)
        ctx.synchronize()
        assert int(r2[0]) == n
        for a, b in zip(ref, small):
            assert np.array_equal(a[:cap], b.cpu().numpy())
        # lower-case n: flagged on the device where the other entry point returns ABW_ERR_LOWER_N
        blob_n, off_n = _assembly(4, 20, lower_n=True)
        t_a = torch.from_numpy(np.concatenate([blob_n, np.zeros((-len(blob_n)) % 16 + 16, np.uint8)])).to(dev)
        t_o = torch.from_numpy(off_n.astype(np.int64)).to(dev)
        signature.segment_static(ctx, t_a, len(blob_n), t_o, 2000, *got_like(dev, 4096), r2)
        ctx.synchronize()
        assert int(r2[1]) != 0
        with pytest.raises(_lib.AbwError) as e:
            signature.segment_dev(ctx, t_a, len(blob_n), t_o, 2000, *got_like(dev, 4096))
        assert e.value.code == _lib.ABW_ERR_LOWER_N
        # beyond the static workspace: refused before anything runs
        L = _lib.lib()
        rc = L.abw_sig_segment_static(ctx.handle, C.c_void_p(t_ascii.data_ptr()), (64 << 20) + 1, C.c_void_p(t_off.data_ptr()), 10, 2000,
                                      0, None, None, None, C.c_void_p(r2.data_ptr()))
        assert rc == _lib.ABW_ERR_CAPACITY and b"static workspace" in L.abw_last_error(ctx.handle)
        rc = L.abw_sig_segment_static(ctx.handle, C.c_void_p(t_ascii.data_ptr()), total, C.c_void_p(t_off.data_ptr()), (256 << 10) + 1, 2000,
                                      0, None, None, None, C.c_void_p(r2.data_ptr()))
        assert rc == _lib.ABW_ERR_CAPACITY
    finally:
        ctx.close()


def got_like(dev, cap):
    import torch
    return [torch.full((cap,), -1, dtype=dt, device=dev) for dt in (torch.int32, torch.int64, torch.int64)]
