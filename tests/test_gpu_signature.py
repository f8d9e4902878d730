"""GPU parity of the signature stage, through the C ABI (abw_sig_build / _segment_dev / _count_dev)
against the committed golden vectors of the compiled reference and against the CPU oracle."""
import os

import numpy as np
import pytest

import _oracle as orc
from abawaca_amd import _lib, signature, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = _lib.Context(0)
    yield c
    c.close()


def _by_name_order(names, seg_scaf):
    order = sorted(range(len(names)), key=lambda i: names[i].encode())
    return np.concatenate([np.flatnonzero(seg_scaf == i) for i in order])


@pytest.mark.parametrize("case", ["sig_edge", "sig_rand"])
def test_build_matches_reference_lrn(ctx, golden_dir, case):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    r = signature.build_signatures(ctx, g["ascii"], g["scaf_off"])
    idx = _by_name_order([str(x) for x in g["scaf_names"]], r["seg_scaf"])
    assert np.array_equal(r["seg_scaf"][idx], g["seg_scaf"].astype(np.uint32))
    assert np.array_equal(r["seg_start"][idx], g["seg_start"])
    assert np.array_equal(r["seg_end"][idx], g["seg_end"])
    assert np.array_equal(r["milli"][idx], g["milli"].astype(np.int32))       # bit-exact vs abawaca.lrn


def _random_assembly(seed, n, maxlen):
    rng = np.random.default_rng(seed)
    seqs = []
    alphabet = np.frombuffer(b"ACGTACGTACGTACGTNRYacgtKM", dtype=np.uint8)
    for i in range(n):
        L = int(rng.integers(0, maxlen))
        s = synth.ACGT[rng.integers(0, 4, size=L)].copy()
        k = int(rng.integers(0, 4))
        for _ in range(k):
            if L > 20:
                a = int(rng.integers(0, L - 1)); b = min(L, a + int(rng.integers(1, 600)))
                s[a:b] = alphabet[rng.integers(0, len(alphabet), size=b - a)]
        seqs.append(s)
    blob = np.concatenate(seqs) if seqs else np.zeros(0, np.uint8)
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    return blob, off


def test_build_matches_oracle_random(ctx):
    blob, off = _random_assembly(3, 120, 12000)
    r = signature.build_signatures(ctx, blob, off)
    seg_scaf, st, en, dims = orc.build_signatures(blob, off)
    assert np.array_equal(r["seg_scaf"], seg_scaf.astype(np.uint32))
    assert np.array_equal(r["seg_start"], st) and np.array_equal(r["seg_end"], en)
    assert np.array_equal(r["dims"].view(np.uint64), dims.view(np.uint64))     # fp64 bit patterns
    milli = np.array([orc.milli_of(x) for x in dims])
    assert np.array_equal(r["milli"], milli[:, 1:])


def test_window_parameter_and_empty_inputs(ctx):
    blob, off = _random_assembly(9, 10, 3000)
    for window in (50, 333, 2000):
        r = signature.build_signatures(ctx, blob, off, window=window)
        exp = []
        for i in range(len(off) - 1):
            rc, s = orc.upcase(blob[int(off[i]):int(off[i + 1])])
            a, b = orc.segments(s, window)
            exp += [(i, int(x), int(y)) for x, y in zip(a, b)]
        got = list(zip(r["seg_scaf"].tolist(), r["seg_start"].tolist(), r["seg_end"].tolist()))
        assert got == exp
    r = signature.build_signatures(ctx, np.zeros(0, np.uint8), np.zeros(1, np.uint64))
    assert len(r["seg_scaf"]) == 0


def test_lowercase_n_is_an_error(ctx):
    blob = np.frombuffer(b"ACGT" * 600 + b"n" + b"ACGT" * 10, dtype=np.uint8)
    with pytest.raises(_lib.AbwError) as e:
        signature.build_signatures(ctx, blob, np.array([0, len(blob)], dtype=np.uint64))
    assert e.value.code == _lib.ABW_ERR_LOWER_N


def test_counts_and_device_entry_points(ctx):
    """abw_sig_segment_dev + abw_sig_count_dev on device tensors, raw counts vs the oracle, and the
    size-independent identities: sum of k-mer counts == number of valid windows; every row of dims
    sums to (number of k with total > 0) within fp rounding."""
    import torch
    blob, off = _random_assembly(21, 300, 9000)
    dev = torch.device("cuda:0")
    pad = (-len(blob)) % 16 + 16
    t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros(pad, np.uint8)])).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    cap = len(blob) // 2000 + len(off) + 16
    t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
    t_st = torch.empty(cap, dtype=torch.int64, device=dev)
    t_en = torch.empty(cap, dtype=torch.int64, device=dev)
    n = signature.segment_dev(ctx, t_ascii, len(blob), t_off, 2000, t_scaf, t_st, t_en)
    h_scaf, h_st, h_en = signature.segment_host(blob, off)
    assert n == len(h_scaf)
    assert np.array_equal(t_scaf[:n].cpu().numpy().astype(np.uint32), h_scaf)
    assert np.array_equal(t_st[:n].cpu().numpy(), h_st) and np.array_equal(t_en[:n].cpu().numpy(), h_en)
    t_dims = torch.empty((n, 180), dtype=torch.float64, device=dev)
    t_counts = torch.empty((n, 340), dtype=torch.int32, device=dev)
    signature.count_dev(ctx, t_ascii, len(blob), t_off, t_scaf, t_st, t_en, n, t_dims=t_dims, t_counts=t_counts)
    ctx.synchronize()
    counts = t_counts.cpu().numpy().astype(np.uint32)
    dims = t_dims.cpu().numpy()
    for s in range(0, n, 7):
        b = int(off[h_scaf[s]]) + int(h_st[s]) - 1
        e = int(off[h_scaf[s]]) + int(h_en[s])
        rc, seg = orc.upcase(blob[b:e])
        _d, c, tot = orc.signature(seg)
        assert np.array_equal(counts[s], c)
        for k, (o, w) in enumerate([(0, 4), (4, 16), (20, 64), (84, 256)]):
            assert counts[s, o:o + w].sum() == tot[k]
    live = np.array([[(counts[s, o:o + w].sum() > 0) for (o, w) in [(0, 4), (4, 16), (20, 64), (84, 256)]] for s in range(n)])
    assert np.allclose(dims.sum(axis=1), live.sum(axis=1), atol=1e-12)


def test_segmentation_with_many_N(ctx):
    """Rank lookup of the segment ends when N is everywhere: sprinkled N (every block has some), long N runs
    (the closing base lies many blocks behind its N-free position), all-N, empty and tiny scaffolds."""
    import torch
    rng = np.random.default_rng(77)
    seqs = []
    for i in range(60):
        L = int(rng.choice([0, 1, 3, 900, 5000, 40000, 150000]))
        s = synth.ACGT[rng.integers(0, 4, size=L)].copy()
        mode = i % 4
        if mode == 0 and L:
            s[rng.random(L) < 0.5] = ord("N")
        elif mode == 1 and L > 100:
            for _ in range(6):
                a = int(rng.integers(0, L - 1)); s[a:a + int(rng.integers(1, 9000))] = ord("N")
        elif mode == 2 and L:
            s[:] = ord("N")
        seqs.append(s)
    blob = np.concatenate(seqs)
    off = np.zeros(len(seqs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    dev = torch.device("cuda:0")
    t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-len(blob)) % 16 + 16, np.uint8)])).to(dev)
    t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
    for window in (2000, 700):
        exp = []
        for i in range(len(seqs)):
            a, b = orc.segments(seqs[i], window)
            exp += [(i, int(x), int(y)) for x, y in zip(a, b)]
        cap = len(exp) + 8
        t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
        t_st = torch.empty(cap, dtype=torch.int64, device=dev)
        t_en = torch.empty(cap, dtype=torch.int64, device=dev)
        n = signature.segment_dev(ctx, t_ascii, len(blob), t_off, window, t_scaf, t_st, t_en)
        assert n == len(exp)
        got = list(zip(t_scaf[:n].cpu().tolist(), t_st[:n].cpu().tolist(), t_en[:n].cpu().tolist()))
        assert got == exp


def test_division_sequence_is_ieee_on_the_whole_domain():
    """The finalisation's hand-scheduled fp64 division (recip_of / quot in abw_sig.hip) against the IEEE division the
    reference performs (src/abawaca-build.cpp:171): EVERY (n, t) a segment below 65 536 bases can produce -- t in
    [1, 65535], n in [0, t], 2.1e9 pairs -- compared on the device with the compiler's own division, and the quotients'
    bit patterns of a few hundred divisors checked against numpy's division on the host; plus sampled dividends for
    divisors up to 2^31 (segments of any length)."""
    import ctypes as C
    ctx = _lib.Context(0)
    try:
        L = _lib.lib()
        bad = C.c_uint64(123)
        sums = np.zeros(65535, dtype=np.uint64)
        ctx.check(L.abw_sig_selftest_division(ctx.handle, 1, 1, 65535, 65535, C.byref(bad), C.c_void_p(sums.ctypes.data)))
        assert bad.value == 0
        for t in list(range(1, 400)) + list(range(401, 65535, 257)) + [65535, 65521, 49999, 32768, 32767]:
            n = np.arange(t + 1, dtype=np.float64)
            want = int((n / float(t)).view(np.uint64).sum(dtype=np.uint64))
            assert int(sums[t - 1]) == want, t
        # larger totals (windows above 65 535 bases): 4097 sampled dividends for each of 8191 divisors up to 2^31
        step = 262139
        cnt = (2 ** 31 - 1 - 65536) // step
        sums2 = np.zeros(cnt, dtype=np.uint64)
        ctx.check(L.abw_sig_selftest_division(ctx.handle, 65536, step, cnt, 4096, C.byref(bad), C.c_void_p(sums2.ctypes.data)))
        assert bad.value == 0
        for i in (0, 1, cnt // 2, cnt - 1):
            t = 65536 + i * step
            n = (np.arange(4097, dtype=np.uint64) * np.uint64(t) // np.uint64(4096)).astype(np.float64)
            assert int(sums2[i]) == int((n / float(t)).view(np.uint64).sum(dtype=np.uint64))
    finally:
        ctx.close()


def test_pipelined_build_over_several_chunks_equals_the_resident_path():
    """abw_sig_build streams host buffers through a ring of pinned staging buffers in chunks of whole scaffolds (48 MB):
    160 Mbp = 4 chunks, with scaffolds of N only in the middle of one (their per-base segments overflow the chunk's
    tables: the synchronous redo path) -- against segmentation + counting of the same blob resident on the device."""
    import torch
    import bench
    blob, off = bench.make_assembly_blob(7, 160)
    blob = blob.copy()
    off = off.astype(np.int64)
    s = 9000                                                      # scaffold 9000 (chunk 2) becomes 10 000 N: 10 000 segments
    blob[off[s]:off[s + 1]] = ord("N")
    total = len(blob)
    ctx = _lib.Context(0)
    try:
        r = signature.build_signatures(ctx, blob, off, want_dims=True, want_milli=True)
        dev = torch.device("cuda", 0)
        t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros((-total) % 16 + 16, np.uint8)])).to(dev)
        t_off = torch.from_numpy(off).to(dev)
        cap = len(r["seg_scaf"]) + 8
        t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
        t_st = torch.empty(cap, dtype=torch.int64, device=dev)
        t_en = torch.empty(cap, dtype=torch.int64, device=dev)
        n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
        assert n == len(r["seg_scaf"]) and n > 85_000
        t_dims = torch.empty((n, 180), dtype=torch.float64, device=dev)
        t_milli = torch.empty((n, 179), dtype=torch.int32, device=dev)
        signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims, t_milli=t_milli)
        ctx.synchronize()
        assert np.array_equal(r["seg_scaf"], t_scaf[:n].cpu().numpy().astype(np.uint32))
        assert np.array_equal(r["seg_start"], t_st[:n].cpu().numpy()) and np.array_equal(r["seg_end"], t_en[:n].cpu().numpy())
        assert np.array_equal(r["dims"].view(np.uint64), t_dims.cpu().numpy().view(np.uint64))
        assert np.array_equal(r["milli"], t_milli.cpu().numpy())
        # a capacity below the count: the count comes back with ABW_ERR_CAPACITY, nothing else is promised
        import ctypes as C
        k = C.c_uint64(0)
        small = np.empty(1000, dtype=np.uint32), np.empty(1000, dtype=np.uint64), np.empty(1000, dtype=np.uint64)
        offu = off.astype(np.uint64)
        rc = _lib.lib().abw_sig_build(ctx.handle, C.c_void_p(blob.ctypes.data), C.c_void_p(offu.ctypes.data), len(off) - 1, 2000, 1000,
                                      C.c_void_p(small[0].ctypes.data), C.c_void_p(small[1].ctypes.data), C.c_void_p(small[2].ctypes.data), None, None, C.byref(k))
        assert rc == _lib.ABW_ERR_CAPACITY and k.value == n
    finally:
        ctx.close()


def test_first_calls_of_two_contexts_from_two_threads():
    """The header promises that different contexts are independent: two host threads whose FIRST library calls run at the same
    moment on two contexts (the process-wide k-mer tables are built and uploaded by whoever comes first).  In a fresh process,
    because these tests have long since made their first call."""
    import subprocess, sys, os
    code = r'''
import sys, threading
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from abawaca_amd import _lib, signature, synth
asm = synth.make_assembly(77, 40, 3, 1, frac_n_run=0.3)
blob = np.concatenate(asm.seqs)
off = np.zeros(len(asm.seqs) + 1, dtype=np.uint64); off[1:] = np.cumsum([len(s) for s in asm.seqs])
out, start = [None, None], threading.Barrier(2)
def work(k):
    start.wait()
    ctx = _lib.Context(0)
    out[k] = signature.build_signatures(ctx, blob, off)
    ctx.close()
th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
[t.start() for t in th]; [t.join() for t in th]
import _oracle as orc
_s, st, en, dims = orc.build_signatures(blob, off)
for r in out:
    assert r is not None and np.array_equal(r["seg_start"], st) and np.array_equal(r["dims"].view(np.uint64), dims.view(np.uint64))
print("ok")
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, "-c", code, root], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.strip().endswith("ok"), p.stderr[-2000:]
