"""Coverage accumulation alone: 100k scaffolds x 10 kbp in 5 kbp segments, reads of 150 bp, unsorted and coordinate-sorted.
usage: python3 experiments/cov_prof.py [million reads]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from abawaca_amd import _lib
dev = torch.device("cuda", 0); torch.zeros(1, device=dev)
n_reads = (int(sys.argv[1]) if len(sys.argv) > 1 else 20) * 1_000_000
S = 100_000
ctx = _lib.Context(0)
seg_scaf = torch.arange(S, device=dev, dtype=torch.int32).repeat_interleave(2)
seg_st = torch.tensor([1, 5001], device=dev, dtype=torch.int64).repeat(S)
seg_en = torch.tensor([5000, 10000], device=dev, dtype=torch.int64).repeat(S)
g = torch.Generator(device=dev); g.manual_seed(1)
r_scaf = torch.randint(0, S, (n_reads,), device=dev, generator=g, dtype=torch.int64)
r_pos = (torch.rand(n_reads, device=dev, generator=g) * (10_000 - 150)).to(torch.int64)
r_len = torch.full((n_reads,), 150, dtype=torch.int32, device=dev)
pv = lambda t: C.c_void_p(t.data_ptr())
def run(rs, rp, label):
    rs32 = rs.to(torch.int32).contiguous(); rp = rp.contiguous()
    cov = torch.zeros(2 * S, dtype=torch.float64, device=dev)
    def step():
        cov.zero_(); torch.cuda.synchronize()
        ctx.check(_lib.lib().abw_cov_accumulate_dev(ctx.handle, pv(seg_scaf), pv(seg_st), pv(seg_en), 2 * S, S, pv(rs32), pv(rp), pv(r_len), n_reads, pv(cov)))
        ctx.synchronize()
    step()
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(3): step()
    n, ms = ctx.prof_read("cov_accumulate"); ctx.prof_enable(False)
    print(f"{label}: {ms / n:.3f} ms per call, {n_reads / (ms / n * 1e-3) / 1e9:.2f} G reads/s, checksum {float(cov.sum()):.6f}", flush=True)
run(r_scaf, r_pos, "unsorted")
key = r_scaf * 20000 + r_pos
order = torch.argsort(key, stable=True)
run(r_scaf[order], r_pos[order], "coordinate-sorted")
