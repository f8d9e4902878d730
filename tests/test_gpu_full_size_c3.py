"""configs[3] at its full size: the signature build of a 10 Gbp assembly (1 000 000 scaffolds of 10 kbp, a run of 50 N in
2 % of them), resident in HBM, checked through size-independent properties computed on the device with torch (the CPU
oracle would need hours; the same properties are checked against numpy at 300 Mbp in test_gpu_full_size.py, where the
oracle-sized cases of test_gpu_signature.py pin the kernels bit for bit):

  segments : per scaffold max(1, nonN // 2000) of them (reference src/abawaca-build.cpp:166-238), consecutive from base 1,
             each holding nonN // nseg non-N bases and ending on the base that fills it
  counts   : the monomer counts of a segment sum to its non-N bases; in a segment without N the k-mer counts sum to
             len - k + 1 (k = 1..4), with N strictly fewer 4-mers; over the assembly the monomer counts sum to the
             bases the segments hold, nseg x (nonN // nseg) per scaffold (a checksum of checksums)
  rows     : the four normalised blocks of every row sum to 4 (|error| <= 1e-12)
  coverage : on those 4.98 M segments (reference src/abawaca-build.cpp:231-244): 10^8 reads of 100 bp that tile every
             scaffold give every segment the coverage 1 (every coordinate of it once; the reference's 0-based read positions
             against 1-based segment coordinates leave a scaffold's last coordinate out); the same reads shifted by 50 bases,
             added to the same sums in a second call, give 2 less the coordinates below 50 (|error| <= 1e-9: the sums are
             fp64 additions of overlap / length in read order)
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import bench  # noqa: E402
from abawaca_amd import _lib, signature  # noqa: E402

pytestmark = pytest.mark.gpu


def test_signature_10gbp_of_configs3():
    import torch
    dev = torch.device("cuda", 0)
    MBP, L = 10_000, 10_000
    t_ascii, total, t_off = bench.make_assembly_blob_gpu(20243, MBP, dev)
    n_scaf = MBP * 100
    assert total == n_scaf * L == 10_000_000_000
    torch.cuda.synchronize()             # torch made the inputs on ITS stream; the library works on its own
    ctx = _lib.Context(0)
    try:
        cap = total // 2000 + n_scaf + 16
        t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
        t_st = torch.empty(cap, dtype=torch.int64, device=dev)
        t_en = torch.empty(cap, dtype=torch.int64, device=dev)
        n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
        t_dims = torch.empty((n, 180), dtype=torch.float64, device=dev)
        t_counts = torch.empty((n, 340), dtype=torch.int32, device=dev)
        signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims, t_counts=t_counts)
        ctx.synchronize()
        scaf = t_scaf[:n].to(torch.int64)
        st, en = t_st[:n], t_en[:n]
        rows = t_ascii[:total].view(n_scaf, L)
        # non-N bases per scaffold, in pieces of 50 000 scaffolds (0.5 GB of text each)
        STEP = 50_000
        nonN = torch.empty(n_scaf, dtype=torch.int64, device=dev)
        for a in range(0, n_scaf, STEP):
            nonN[a:a + STEP] = (rows[a:a + STEP] != ord("N")).sum(1)
        nseg = torch.clamp(nonN // 2000, min=1)
        assert n == int(nseg.sum())
        assert torch.equal(torch.bincount(scaf, minlength=n_scaf), nseg)
        first = torch.cumsum(nseg, 0) - nseg
        assert bool((st[first] == 1).all())
        same = scaf[1:] == scaf[:-1]
        assert bool((scaf[1:] >= scaf[:-1]).all()) and bool((st[1:][same] == en[:-1][same] + 1).all())
        assert bool((en >= st).all()) and bool((en <= L).all())
        # per segment: its non-N bases from a per-scaffold prefix sum, the base it ends on
        want = (nonN // nseg)[scaf]
        seg_nonN = torch.empty(n, dtype=torch.int64, device=dev)
        ends_on_base = torch.empty(n, dtype=torch.bool, device=dev)
        for a in range(0, n_scaf, STEP):
            b = min(n_scaf, a + STEP)
            s0, s1 = int(first[a]), int(first[b - 1] + nseg[b - 1])
            isb = rows[a:b] != ord("N")
            cum = torch.zeros((b - a, L + 1), dtype=torch.int32, device=dev)
            cum[:, 1:] = torch.cumsum(isb, 1, dtype=torch.int32)
            ls = scaf[s0:s1] - a
            seg_nonN[s0:s1] = (cum[ls, en[s0:s1]] - cum[ls, st[s0:s1] - 1]).to(torch.int64)
            ends_on_base[s0:s1] = isb[ls, en[s0:s1] - 1]
            del isb, cum
        assert torch.equal(seg_nonN, want)
        assert bool(ends_on_base.all())
        # counts
        length = en - st + 1
        clean = seg_nonN == length
        tot = [t_counts[:, o:o + w].sum(1, dtype=torch.int64) for o, w in ((0, 4), (4, 16), (20, 64), (84, 256))]
        assert torch.equal(tot[0], seg_nonN)
        assert int(tot[0].sum()) == int((nseg * (nonN // nseg)).sum())   # the checksum of checksums: what the segments hold
        assert 0 <= int(nonN.sum()) - int(tot[0].sum()) < 4 * n_scaf     # (the reference drops nonN mod nseg bases per scaffold)
        for k in range(1, 5):
            assert torch.equal(tot[k - 1][clean], (length - k + 1)[clean])
        n_dirty = int((~clean).sum())                                   # (a run in a scaffold's dropped tail is in no segment)
        assert n_scaf // 100 <= n_dirty <= 2 * (n_scaf // 50)
        assert bool((tot[3][~clean] < (length - 3)[~clean]).all())
        # rows: four normalised blocks
        err = (t_dims.sum(1) - 4.0).abs().max()
        assert float(err) <= 1e-12
        del t_dims, t_counts, tot, seg_nonN, ends_on_base, want
        # coverage of the same segments
        RL = 100
        n_reads = total // RL
        i = torch.arange(n_reads, device=dev, dtype=torch.int64)
        r_scaf = (i // (L // RL)).to(torch.int32)
        r_pos = (i % (L // RL)) * RL
        r_len = torch.full((n_reads,), RL, dtype=torch.int32, device=dev)
        del i
        t_cov = torch.zeros(n, dtype=torch.float64, device=dev)
        pv = lambda t: C.c_void_p(t.data_ptr())
        cov_call = lambda pos: ctx.check(_lib.lib().abw_cov_accumulate_dev(ctx.handle, pv(t_scaf), pv(t_st), pv(t_en), n, n_scaf, pv(r_scaf), pv(pos),
                                                                          pv(r_len), n_reads, pv(t_cov)))
        # The reference compares a read's 0-based position with the segments' 1-based coordinates as they are (ref
        # src/ReadMapping.cpp:41, src/abawaca-build.cpp:177-185, 231-244): the tiling reads cover the coordinates 0 .. L - 1,
        # so a segment that ends on coordinate L misses one base; the shifted reads cover 50 .. L + 49.
        flen = length.to(torch.float64)
        torch.cuda.synchronize()
        cov_call(r_pos)
        ctx.synchronize()
        expect = 1.0 - (en == L).to(torch.float64) / flen
        assert float((t_cov - expect).abs().max()) <= 1e-9
        assert 0 < int((en == L).sum()) <= n_scaf
        r_pos2 = r_pos + RL // 2
        torch.cuda.synchronize()
        cov_call(r_pos2)                                                # continues the sums
        ctx.synchronize()
        missing = torch.clamp(RL // 2 - st, min=0).to(torch.float64)    # coordinates st .. 49
        expect = expect + 1.0 - missing / flen
        assert float((t_cov - expect).abs().max()) <= 1e-9
        assert int((missing > 0).sum()) == n_scaf
    finally:
        ctx.close()
