"""Does the signature stage gain from running the segmentation of one part of the assembly beside the counting of another?
1 Gbp resident; (a) one context, segmentation then counting over the whole blob (the bench's way); (b) the blob cut in P parts
at scaffold boundaries, two contexts (two streams) driven by two host threads, each takes every other part: while one counts,
the other segments.  Same rows (checksum)."""
import os, sys, time, threading
sys.path.insert(0, "/root/repo")
import numpy as np, torch, bench
from abawaca_amd import _lib, signature
dev = torch.device("cuda", 0)
blob, off = bench.make_assembly_blob(20244, 1000)
total = len(blob); pad = (-total) % 16 + 16
t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros(pad, np.uint8)])).to(dev)
off = off.astype(np.int64)
t_off = torch.from_numpy(off).to(dev)
cap = total // 2000 + len(off) + 16
def bufs(c):
    return (torch.empty(c, dtype=torch.int32, device=dev), torch.empty(c, dtype=torch.int64, device=dev), torch.empty(c, dtype=torch.int64, device=dev),
            torch.empty((c, 180), dtype=torch.float64, device=dev))
ctx = _lib.Context(0)
sc, st, en, dims = bufs(cap)
def whole():
    n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, sc, st, en)
    signature.count_dev(ctx, t_ascii, total, t_off, sc, st, en, n, t_dims=dims)
    ctx.synchronize()
    return n
n = whole()
t0 = time.perf_counter()
for _ in range(5): whole()
dt_a = (time.perf_counter() - t0) / 5
chk_a = dims[:n].view(torch.int64).sum().item()
print(f"one context: {dt_a * 1e3:.3f} ms wall per pass, {total / dt_a / 1e9:.0f} Gbp/s, {n} segments", flush=True)
for P in (2, 4, 8):
    n_scaf = len(off) - 1
    cuts = [0] + [int(np.searchsorted(off, total * k // P)) for k in range(1, P)] + [n_scaf]
    parts = []
    for k in range(P):
        a, b = cuts[k], cuts[k + 1]
        lo = int(off[a]) & ~15                                  # 16-byte aligned start of the part's view
        view = t_ascii[lo:]
        po = torch.from_numpy(off[a:b + 1] - lo).to(dev)
        plen = int(off[b]) - lo
        pc = plen // 2000 + (b - a) + 16
        parts.append((view, plen, po, bufs(pc)))
    ctxs = [_lib.Context(0), _lib.Context(0)]
    res = [0] * P
    def work(w):
        c = ctxs[w]
        for k in range(w, P, 2):
            view, plen, po, (s1, s2, s3, d) = parts[k]
            m = signature.segment_dev(c, view, plen, po, 2000, s1, s2, s3)
            signature.count_dev(c, view, plen, po, s1, s2, s3, m, t_dims=d)
            res[k] = m
        c.synchronize()
    def both():
        th = [threading.Thread(target=work, args=(w,)) for w in (0, 1)]
        for t in th: t.start()
        for t in th: t.join()
    both()
    t0 = time.perf_counter()
    for _ in range(5): both()
    dt_b = (time.perf_counter() - t0) / 5
    chk_b = sum(parts[k][3][3][:res[k]].view(torch.int64).sum().item() for k in range(P))
    chk_b = (chk_b + 2 ** 63) % 2 ** 64 - 2 ** 63
    print(f"two contexts, {P} parts: {dt_b * 1e3:.3f} ms wall per pass, {total / dt_b / 1e9:.0f} Gbp/s, {sum(res)} segments, same rows: {chk_b == chk_a}", flush=True)
    del parts, ctxs
