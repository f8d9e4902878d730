#!/usr/bin/env python3
"""bench.py — throughput of the abawaca hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Headline (BASELINE.json "metric"): k-means point·iterations/s over the whole node on the workload
of configs[1] (200k scaffolds -> ~0.8 M dps, 179 signature + 10 coverage dims, 64 restarts), plus
the signature build in Gbp/s and the real separator in dimension·points/s as extra keys of the same
JSON line.  A "step" is one full pass: one k-means run of `--iters` Lloyd iterations over all
restarts, inputs already resident in HBM.  With N GPUs every rank runs its own 64 restarts on a
replica of the points (the north-star's "restarts spread across GPUs"): weak scaling, no data-path
collective; value = units of all ranks / max time over ranks.

The JSON line also carries
  roofline     : the dominant kernel (km_step) against the dense fp32 MFMA peak, its duration taken
                 live from HIP events recorded by the library on its own stream (abw_prof_*)
  cpu_baseline : the CPU restatement of the same k-means (oracle/, "port" — the reference has no
                 k-means) on a bounded sample, on this box's host cores
and, as secondary objects, the reference's own binaries (oracle/_ref, "reference") timed on
bounded samples of the signature and separator workloads.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scaffolds", type=int, default=200_000)
    ap.add_argument("--samples", type=int, default=10)
    ap.add_argument("--restarts", type=int, default=64)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--chunk", type=int, default=0, help="points per fp32 partial sum; 0 = the library's choice (abw_km_auto_chunk)")
    ap.add_argument("--sig-mbp", type=int, default=1000, help="signature side bench: assembly size in Mbp")
    ap.add_argument("--cov-mreads", type=int, default=20, help="coverage side bench: million reads of 150 bp")
    ap.add_argument("--no-extras", action="store_true", help="only the k-means headline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1: nccl (= RCCL over xGMI); "
                    "gloo lets several ranks share one GPU for rehearsals (collectives staged through the host)")
    ap.add_argument("--no-sep-sharded", action="store_true", help="skip the scaffold-sharded separator leg of a multi-rank run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines")
    ap.add_argument("--c1", action="store_true", help="also run configs[0] end to end: the whole recursive binning of a 10k-scaffold "
                    "table, reference driver (its own separator, all host cores; takes about a minute) beside the GPU separator")
    return ap.parse_args()


def make_points(seed, n_scaf, n_samples, device):
    """configs[1]-shaped feature table generated on the GPU (same model as abawaca_amd.synth.make_features:
    per-genome GC -> expected canonical k-mer frequencies + sampling noise, truncated to thousandths;
    coverage c_{g,s}(1+N(0,0.1))).  -> (keys int32 [D, N] on device, dp2scaf, scaf_len, genome)"""
    import torch
    from abawaca_amd import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    G = max(2, n_scaf // 2000)
    rng = synth.rng_for(seed)
    mer2dim, _names = synth.canonical_mer_table()
    gc = np.linspace(0.30, 0.70, G)
    cov = np.exp(rng.normal(np.log(20.0), 1.0, size=(G, n_samples)))
    genome = rng.integers(0, G, size=n_scaf).astype(np.int32)
    ndps = rng.integers(2, 7, size=n_scaf).astype(np.int32)
    scaf_len = (2000 * ndps + rng.integers(0, 2000, size=n_scaf)).astype(np.int64)
    dp2scaf = np.repeat(np.arange(n_scaf, dtype=np.int32), ndps)
    N = int(ndps.sum())
    p_gc_s = np.clip(gc[genome] + rng.normal(0, 0.004, size=n_scaf), 0.05, 0.95)
    p = torch.from_numpy(p_gc_s[dp2scaf]).to(device)
    D = 179 + n_samples
    keys = torch.empty((D, N), dtype=torch.int32, device=device)
    col_terms = [[] for _ in range(180)]
    off = 0
    for k in range(1, 5):
        for m in range(4 ** k):
            n_gc = sum(1 for i in range(k) if ((m >> (2 * i)) & 3) in (1, 2))
            col_terms[mer2dim[off + m]].append((n_gc, k - n_gc))
        off += 4 ** k
    for c in range(1, 180):
        f = torch.zeros(N, dtype=torch.float64, device=device)
        for n_gc, n_at in col_terms[c]:
            f += (p / 2) ** n_gc * ((1 - p) / 2) ** n_at
        noise = torch.randn(N, dtype=torch.float64, device=device, generator=g) * torch.sqrt(torch.clamp(f * (1 - f), min=0) / 2500.0)
        keys[c - 1] = torch.clamp(torch.floor((f + noise) * 1000.0), 0, 1000).to(torch.int32)
    gi = torch.from_numpy(genome[dp2scaf].astype(np.int64)).to(device)
    for s in range(n_samples):
        cs = torch.from_numpy(cov[:, s]).to(device)[gi] * (1.0 + 0.1 * torch.randn(N, dtype=torch.float64, device=device, generator=g))
        keys[179 + s] = torch.floor(torch.clamp(cs, min=0) * 1000.0).to(torch.int32)
    return keys, dp2scaf, scaf_len, genome, G


def standardise(keys_t):
    """k-means features: every dimension centred and scaled to unit variance (fp32, [D, N])."""
    import torch
    x = keys_t.to(torch.float32)
    mu = x.mean(dim=1, keepdim=True)
    sd = x.std(dim=1, keepdim=True).clamp_min(1e-6)
    return ((x - mu) / sd).contiguous()


def make_assembly_blob(seed, mbp):
    """configs[3]-shaped assembly: 10 kbp scaffolds, iid bases with per-scaffold GC, 50-N runs in 2 %."""
    rng = np.random.Generator(np.random.PCG64(seed))
    n_scaf = mbp * 100
    L = 10_000
    codes = rng.integers(0, 256, size=n_scaf * L // 4, dtype=np.uint8)      # 4 bases per random byte
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    blob = np.empty(n_scaf * L, dtype=np.uint8)
    for sh in range(4):
        blob[sh::4] = lut[(codes >> (2 * sh)) & 3]
    for s in rng.choice(n_scaf, size=max(1, n_scaf // 50), replace=False):
        at = int(s) * L + int(rng.integers(0, L - 50))
        blob[at:at + 50] = ord("N")
    off = (np.arange(n_scaf + 1, dtype=np.uint64) * L)
    return blob, off


def cpu_kmeans_baseline(x_host, R, iters, chunk):
    import _oracle as orc
    cores = min(16, os.cpu_count() or 1)
    d, N = x_host.shape
    # calibrate on a small slice, then size the sample (points first, then iterations) for ~12 s
    n0 = min(N, 4000)
    t0 = time.perf_counter()
    orc.kmeans(x_host[:, :n0].copy(), cores, 1, 1, chunk, nthreads=cores)
    dt = max(time.perf_counter() - t0, 1e-3)
    rate = n0 * cores * 2 / dt                         # point*restart passes per second (1 iteration + assign pass)
    Rs = cores * 4
    n1 = int(min(N, max(n0, rate * 12.0 / (Rs * 3))))
    its = int(min(40, max(2, round(rate * 12.0 / (n1 * Rs)) - 1)))
    t0 = time.perf_counter()
    orc.kmeans(x_host[:, :n1].copy(), Rs, 1, its, chunk, nthreads=cores)
    dt = time.perf_counter() - t0
    return dict(value=n1 * Rs * its / dt, unit="point*iterations/s", cores=cores, kind="port",
                sample=f"oracle abw_or_kmeans (OpenMP, {cores} threads): first {n1} points x {Rs} restarts x {its} iterations "
                       f"(+1 assign pass, not counted) in {dt:.1f} s; the reference has no k-means")


def cpu_signature_baseline(blob, off, mbp_sample=200):
    exe = os.path.join(ROOT, "oracle", "_ref", "abawaca-build")
    if not os.path.exists(exe):
        return None
    n_scaf = min(len(off) - 1, mbp_sample * 100)
    end = int(off[n_scaf])
    with tempfile.TemporaryDirectory(prefix="abwbench_") as wd:
        fa = os.path.join(wd, "a.fa")
        with open(fa, "wb") as f:
            for i in range(n_scaf):
                f.write(b">s%d\n" % i)
                f.write(blob[int(off[i]):int(off[i + 1])].tobytes())
                f.write(b"\n")
        sam = os.path.join(wd, "s0.sam")
        with open(sam, "w") as f:
            f.write("@HD\tVN:1.0\nr0\t0\ts0\t1\t255\t50M\t*\t0\t0\t" + blob[:50].tobytes().decode() + "\t" + "I" * 50 + "\n")
        out = os.path.join(wd, "out")
        os.makedirs(out)
        t0 = time.perf_counter()
        subprocess.run([exe, "-f", fa, "-o", out, "-s", sam, "-c", sam], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.perf_counter() - t0
    return dict(value=end / dt / 1e9, unit="Gbp/s", cores=1, kind="reference",
                sample=f"reference abawaca-build (-O2 build of its own sources; the program is single-threaded) on the first "
                       f"{end / 1e6:.0f} Mbp of the same assembly, whole program incl. FASTA read and .lrn write, {dt:.1f} s")


def cpu_separator_baseline(keys_host, dp2scaf, scaf_len, n_scaf_sample=12000):
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_sep_driver")
    if not os.path.exists(exe):
        return None
    S = min(n_scaf_sample, len(scaf_len))
    n = int(np.searchsorted(dp2scaf, S))
    D = keys_host.shape[0]
    cores = min(40, os.cpu_count() or 1)                 # the reference caps -p at 40 (src/abawaca.cpp:328)
    with tempfile.TemporaryDirectory(prefix="abwbench_") as wd:
        with open(os.path.join(wd, "a.fa"), "w") as f:
            for s in range(S):
                f.write(f">s{s:07d}\n" + "A" * int(scaf_len[s]) + "\n")
        with open(os.path.join(wd, "a.info"), "w") as f:
            for s in range(S):
                f.write(f"s{s:07d}\t{int(scaf_len[s])}\t1.000\t0.500\t0\n")
        with open(os.path.join(wd, "a.names"), "w") as f:
            f.write(f"% {n}\n")
            for p in range(n):
                s = int(dp2scaf[p])
                f.write(f"{p + 1}\ts{s:07d}_{p}\ts{s:07d}:(1, 2000), 2000/2000 non-Ns bps\n")
        with open(os.path.join(wd, "a.lrn"), "w") as f:
            f.write(f"% {n}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\td{j}" for j in range(D)) + "\n")
            import pandas as pd
            df = pd.DataFrame(keys_host[:, :n].T / 1000.0, index=np.arange(1, n + 1))
            df.to_csv(f, sep="\t", header=False, float_format="%.3f", lineterminator="\n")
        open(os.path.join(wd, "scg.list"), "w").write("SCG00\n")
        prefix = os.path.join(wd, "ref")
        subprocess.run([exe, os.path.join(wd, "a.names"), os.path.join(wd, "a.fa"), os.path.join(wd, "a.info"),
                        os.path.join(wd, "a.lrn"), "-", os.path.join(wd, "scg.list"), str(cores), prefix, "1"],
                       check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        row = np.loadtxt(prefix + ".nodes.tsv", ndmin=2)[0]
    usec = float(row[11])
    return dict(value=D * n / (usec * 1e-6), unit="dimension*points/s", cores=cores, kind="reference",
                sample=f"reference ClusterSeparatorBySensitivitySpecificity::separate() (-O2 build of its own sources, "
                       f"-p {cores}) on the root node of the first {S} scaffolds ({n} dps x {D} dims), {usec / 1e6:.2f} s")


def lrn_bench(ctx, with_cpu):
    """row N4 (next), loader side: the .lrn text of a configs[1]-shaped table (200 000 rows x 189 columns of the
    bench table's widths, fixed-width so that numpy can write it quickly) -> int32 keys."""
    from abawaca_amd import lrn
    rng = np.random.default_rng(20248)
    N, Ds, Dc = 200_000, 179, 10
    sig = rng.integers(0, 1001, size=(N, Ds), dtype=np.int64)
    cov = rng.integers(0, 10_000_000, size=(N, Dc), dtype=np.int64)
    row_len = 7 + Ds * 6 + Dc * 9 + 1
    buf = np.empty((N, row_len), dtype=np.uint8)
    names = np.arange(1, N + 1)
    for k in range(7):
        buf[:, 6 - k] = 48 + (names // 10 ** k) % 10
    o = 7
    s3 = buf[:, o:o + Ds * 6].reshape(N, Ds, 6)
    s3[:, :, 0] = 9; s3[:, :, 1] = 48 + sig // 1000; s3[:, :, 2] = 46
    for k in range(3):
        s3[:, :, 5 - k] = 48 + (sig // 10 ** k) % 10
    o += Ds * 6
    c3 = buf[:, o:o + Dc * 9].reshape(N, Dc, 9)
    c3[:, :, 0] = 9; c3[:, :, 5] = 46
    for k in range(3):
        c3[:, :, 8 - k] = 48 + (cov // 10 ** k) % 10
    for k in range(4):
        c3[:, :, 4 - k] = 48 + (cov // 10 ** (3 + k)) % 10
    buf[:, -1] = 10
    D = Ds + Dc
    head = (f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\tc{j}" for j in range(D)) + "\n").encode()
    text = np.concatenate([np.frombuffer(head, np.uint8), buf.reshape(-1)])
    dp, keys, _ = lrn.lrn_parse(ctx, text)                       # warm-up + check
    assert np.array_equal(dp, names) and np.array_equal(keys[:Ds], sig.T) and np.array_equal(keys[Ds:], cov.T)
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        lrn.lrn_parse(ctx, text)
    wall = (time.perf_counter() - t0) / reps
    n, ms = ctx.prof_read("lrn_parse")
    ctx.prof_enable(False)
    k_s = ms / reps * 1e-3
    alg = 2.0 * len(text) + 4.0 * D * N                          # the text twice (lines, then fields) + the keys
    res = dict(value=len(text) / k_s / 1e9, unit="GB/s of text (kernels, text resident)", rows=N, columns=D, text_mb=round(len(text) / 1e6, 1),
               ms_kernels=round(k_s * 1e3, 3), ms_with_pcie_both_ways=round(wall * 1e3, 2), fields_per_s=round(N * (D + 1) / k_s / 1e9, 2),
               note="row N4 (next), loader side; fields_per_s in G fields/s; wall includes H2D of the text and D2H of the keys from pageable memory",
               roofline=dict(kernel="lrn_rows_kernel (+ line scan)", bound="hbm", achieved=round(alg / k_s / 1e9, 1), peak=HBM_PEAK_GBS,
                             unit="GB/s", frac=round(alg / k_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None))
    # writer side: the same table back to text (reference src/abawaca-build.cpp:595-607), checked by parsing it again
    rows_m = np.ascontiguousarray(np.concatenate([sig, cov], axis=1).astype(np.int32))
    out_text = lrn.lrn_emit(ctx, rows_m, 1)
    dp2, keys2, _ = lrn.lrn_parse(ctx, np.concatenate([np.frombuffer(head, np.uint8), out_text]))
    assert np.array_equal(dp2, names) and np.array_equal(keys2.T, rows_m)
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(reps):
        lrn.lrn_emit(ctx, rows_m, 1)
    wall_e = (time.perf_counter() - t0) / reps
    ne, mse = ctx.prof_read("lrn_emit")
    ctx.prof_enable(False)
    e_s = mse / reps * 1e-3
    res["writer"] = dict(value=len(out_text) / e_s / 1e9, unit="GB/s of text (kernels)", text_mb=round(len(out_text) / 1e6, 1),
                         ms_kernels=round(e_s * 1e3, 3), ms_with_pcie_both_ways=round(wall_e * 1e3, 2),
                         roofline=dict(kernel="lrn_emit_kernel (+ row lengths)", bound="hbm", achieved=round((8.0 * D * N + len(out_text)) / e_s / 1e9, 1),
                                       peak=HBM_PEAK_GBS, unit="GB/s", frac=round((8.0 * D * N + len(out_text)) / e_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None))
    # cluster writer (reference src/ClusterWriter.cpp:92-98): the table as resident doubles, 100 clusters of 2 000 datapoints
    # each written as "<dp>\t%lf..." rows, one call per cluster
    import torch
    t_vals = torch.from_numpy(np.ascontiguousarray(rows_m.T.astype(np.float64) / 1000.0)).cuda()     # [D][N]
    perm = torch.from_numpy(rng.permutation(N).astype(np.int64) + 1).cuda()
    n_cl = 100
    per = N // n_cl
    t_out = torch.empty(per * (D * 14 + 12), dtype=torch.uint8, device="cuda")
    def clusters():
        tot = 0
        for c in range(n_cl):
            tot += lrn.lrn_emit_rows_dev(ctx, t_vals, perm[c * per:(c + 1) * per], t_out)
        return tot
    clusters()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    cbytes = clusters()
    wall_c = time.perf_counter() - t0
    nc, msc = ctx.prof_read("lrn_emit_rows")
    ctx.prof_enable(False)
    c_s = msc * 1e-3
    res["cluster_writer"] = dict(value=cbytes / c_s / 1e9, unit="GB/s of text (kernels)", clusters=n_cl, rows_per_cluster=per, text_mb=round(cbytes / 1e6, 1),
                                 ms_kernels=round(c_s * 1e3, 3), ms_wall_text_left_on_device=round(wall_c * 1e3, 2),
                                 roofline=dict(kernel="lrn_emit6_kernel (+ row lengths)", bound="hbm", achieved=round((16.0 * D * N + cbytes) / c_s / 1e9, 1),
                                               peak=HBM_PEAK_GBS, unit="GB/s", frac=round((16.0 * D * N + cbytes) / c_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                                               note="algorithmic bytes: every value read twice (length pass, print pass) + the text once"))
    # the same clusters in ONE call (concatenated lists, row offsets back): no longer bound by five launches per cluster
    t_all = torch.empty(cbytes, dtype=torch.uint8, device="cuda")
    t_off = torch.empty(n_cl * per + 1, dtype=torch.int64, device="cuda")
    lrn.lrn_emit_rows_dev(ctx, t_vals, perm[:n_cl * per], t_all, t_off)
    ctx.prof_reset(); ctx.prof_enable(True)
    for _ in range(reps):
        lrn.lrn_emit_rows_dev(ctx, t_vals, perm[:n_cl * per], t_all, t_off)
    no, mso = ctx.prof_read("lrn_emit_rows")
    ctx.prof_enable(False)
    o_s = mso / reps * 1e-3
    res["cluster_writer"]["one_call"] = dict(value=cbytes / o_s / 1e9, unit="GB/s of text (kernels)", ms_kernels=round(o_s * 1e3, 3),
                                             roofline_frac=round((16.0 * D * N + cbytes) / o_s / 1e9 / HBM_PEAK_GBS, 4))
    del t_all, t_off
    if with_cpu:
        import _oracle as orc
        hv = t_vals.cpu().numpy()
        hd = perm[:per].cpu().numpy()
        t0 = time.perf_counter()
        ref_rows = orc.lrn_emit_rows(hv, hd)
        dt = time.perf_counter() - t0
        chk = torch.empty(len(ref_rows), dtype=torch.uint8, device="cuda")
        lrn.lrn_emit_rows_dev(ctx, t_vals, perm[:per], chk)
        assert np.array_equal(chk.cpu().numpy(), ref_rows)
        res["cluster_writer"]["cpu_baseline"] = dict(value=len(ref_rows) / dt / 1e9, unit="GB/s of text", cores=1, kind="port",
                                                     sample=f"oracle abw_or_lrn_emit_rows (the reference's fprintf loop into memory), one cluster of {per} rows, {dt:.2f} s")
        del chk
    del t_vals, t_out
    if with_cpu:
        import _oracle as orc
        ns = 20_000
        sample = np.concatenate([np.frombuffer(head.replace(f"% {N}".encode(), f"% {ns}".encode(), 1), np.uint8), buf[:ns].reshape(-1)])
        t0 = time.perf_counter()
        orc.lrn_parse(sample, ns + 1, D)
        dt = time.perf_counter() - t0
        res["cpu_baseline"] = dict(value=len(sample) / dt / 1e9, unit="GB/s of text", cores=1, kind="port",
                                   sample=f"oracle abw_or_lrn_parse (the reference's fgets/strchr/atof loop, without its std::map and "
                                          f"Dimension inserts), first {ns} rows, {dt:.2f} s")
    return res


def recursion_c1(ctx):
    """configs[0]: 10 000 scaffolds, 179 signature + 3 coverage columns, the whole recursion (reference
    src/abawaca.cpp:95-197) -- reference objects + reference separator (oracle/_ref/ref_sep_driver, -O2) beside
    abawaca_amd.separator.binning on the GPU; the two must give the same tree and the same bins."""
    from abawaca_amd import separator, synth
    f = synth.make_features(20240, 10000, 25, 3)
    S = len(f["scaf_len"])
    off, ids = separator.scg_csr(f["scg_scaf"], f["scg_id"], S)
    t0 = time.perf_counter()
    store = separator.ColumnStore(ctx, f["milli"], f["dp2scaf"], f["scaf_len"], off, ids, f["n_scg"])
    load_s = time.perf_counter() - t0
    separator.binning(store, max_nodes=3)                     # warm-up
    t0 = time.perf_counter()
    nodes, s2c, d2c = separator.binning(store)
    gpu_s = time.perf_counter() - t0
    res = dict(workload=f"configs[0]: {S} scaffolds -> {f['milli'].shape[1]} dps x {f['milli'].shape[0]} dims, {f['n_scg']} SCGs, whole recursion",
               nodes=len(nodes), bins=int(sum(1 for r in nodes if not r["split"])), gpu_s=round(gpu_s, 4), gpu_load_s=round(load_s, 3),
               gpu_ms_per_node=round(gpu_s / len(nodes) * 1e3, 3))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_sep_driver")
    if os.path.exists(exe):
        cores = min(40, os.cpu_count() or 1)
        with tempfile.TemporaryDirectory(prefix="abwc1_") as wd:
            files = synth.write_build_dir(wd, f["milli"], f["dp2scaf"], f["scaf_len"], f["scg_scaf"], f["scg_id"], f["n_scg"])
            t0 = time.perf_counter()
            subprocess.run([exe, files["names"], files["fasta"], files["info"], files["lrn"], files["scg"], files["scg_list"],
                            str(cores), os.path.join(wd, "ref")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            wall = time.perf_counter() - t0
            # the same driver and reference objects with integration/ClusterSeparatorHip in place of the reference's separator:
            # what a user of the reference sees after the one-line change (adapter work on the reference's containers included)
            hip_exe = os.path.join(ROOT, "oracle", "_ref", "hip_sep_driver")
            if os.path.exists(hip_exe):
                t0 = time.perf_counter()
                subprocess.run([hip_exe, files["names"], files["fasta"], files["info"], files["lrn"], files["scg"], files["scg_list"],
                                str(cores), os.path.join(wd, "hip")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                hip_wall = time.perf_counter() - t0
                hrows = np.loadtxt(os.path.join(wd, "hip.nodes.tsv"), ndmin=2)
                res["dropin"] = dict(separate_s=round(float(hrows[:, 11].sum()) / 1e6, 3), first_node_s=round(float(hrows[0, 11]) / 1e6, 3),
                                     wall_s=round(hip_wall, 2), identical_output=bool(
                                         open(os.path.join(wd, "hip.scaf2cluster.tsv")).read() == open(os.path.join(wd, "ref.scaf2cluster.tsv")).read()
                                         and np.array_equal(hrows[:, :11], np.loadtxt(os.path.join(wd, "ref.nodes.tsv"), ndmin=2)[:, :11])),
                                     note="reference main loop + reference containers + ClusterSeparatorHip (C++ adapter over the C ABI); "
                                          "first_node_s includes building the column store from the reference's Dimension objects and abw_sep_load")
            # the same two drivers writing every cluster's files (reference src/abawaca.cpp:161-182): the reference's
            # ClusterEsomWriter (fprintf per value) beside integration/ClusterEsomWriterHip (rows printed on the GPU)
            if os.path.exists(hip_exe):
                wr = {}
                for tag, e in (("ref", exe), ("hip", hip_exe)):
                    cdir = os.path.join(wd, "clusters_" + tag)
                    os.makedirs(cdir)
                    t0 = time.perf_counter()
                    subprocess.run([e, files["names"], files["fasta"], files["info"], files["lrn"], files["scg"], files["scg_list"],
                                    str(cores), os.path.join(wd, tag + "w")], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                   env=dict(os.environ, ABW_CLUSTER_DIR=cdir))
                    w = time.perf_counter() - t0
                    us, n = open(os.path.join(wd, tag + "w.writer.tsv")).read().split()
                    wr[tag] = dict(wall_s=round(w, 2), writer_s=round(int(us) / 1e6, 3), clusters=int(n),
                                   bytes=sum(os.path.getsize(os.path.join(cdir, x)) for x in os.listdir(cdir)))
                same = all(open(os.path.join(wd, "clusters_ref", x), "rb").read() == open(os.path.join(wd, "clusters_hip", x), "rb").read()
                           for x in os.listdir(os.path.join(wd, "clusters_ref")))
                res["with_cluster_files"] = dict(reference=wr["ref"], dropin=wr["hip"], identical_files=bool(same and wr["ref"]["bytes"] == wr["hip"]["bytes"]),
                                                 note="writer_s: time inside ClusterWriter::write for all clusters (+ the subset tables the "
                                                      "reference builds for it); the GPU writer's first call also uploads the root table")
            rows = np.loadtxt(os.path.join(wd, "ref.nodes.tsv"), ndmin=2)
            ref_s2c = np.array([int(ln.split("\t")[1]) for ln in open(os.path.join(wd, "ref.scaf2cluster.tsv"))], dtype=np.int32)
        same_tree = len(rows) == len(nodes) and all(
            (int(r[0]), int(r[1]), int(r[2]), int(r[3])) == (a["id"], a["ndps"], a["split"], a["dim"]) and
            (a["dim"] <= 0 or a["key"] / 1000.0 == r[4]) and a["spec"] == r[5] and a["sens"] == r[6] and
            (int(r[7]), int(r[8]), int(r[9]), int(r[10])) == (a["c1"], a["n1"], a["c2"], a["n2"]) for r, a in zip(rows, nodes))
        res["reference"] = dict(separate_s=round(float(rows[:, 11].sum()) / 1e6, 2), wall_s=round(wall, 2), cores=cores,
                                note="reference ClusterSeparatorBySensitivitySpecificity (-O2 build of its own sources) inside our driver of "
                                     "its main loop; separate_s = sum of its separate() calls, wall_s also has file loading and the "
                                     "per-node ClusterData subsets")
        res["identical_tree"] = bool(same_tree)
        res["identical_bins"] = bool(np.array_equal(ref_s2c, s2c))
        res["speedup_vs_separate_s"] = round(float(rows[:, 11].sum()) / 1e6 / gpu_s, 1)
    return res


def main():
    args = parse()
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_index = local_rank % max(1, torch.cuda.device_count()) if args.backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        dist = dist_mod
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from abawaca_amd import _lib, kmeans, multi, separator, signature
    ctx = _lib.Context(dev_index)

    # ---------------- workload: configs[1] -------------------------------------------------------
    keys_t, dp2scaf, scaf_len, genome, G = make_points(20241, args.scaffolds, args.samples, dev)
    D, N = keys_t.shape
    x_t = standardise(keys_t)
    R, iters = args.restarts, args.iters
    if args.chunk == 0:
        args.chunk = kmeans.auto_chunk(ctx, N, R)   # what the library itself picks for chunk = 0
    # weak scaling: the job is world*R restarts, rank k owns restart ids [k*R, (k+1)*R) (abawaca_amd/multi.py)
    first_restart, _n = multi.restart_range(world * R, rank, world)
    seed = multi.rank_seed(20240, first_restart)
    t_inertia = torch.empty(R, dtype=torch.float64, device=dev)
    t_cent = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
    t_assign = torch.empty((R, N), dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    def step():
        kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R, seed, iters, args.chunk, t_assign, t_cent, t_inertia)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_launch, ms_launch = ctx.prof_read("km_step")
    ctx.prof_enable(False)
    if dist is not None:
        elapsed = multi.max_elapsed(elapsed, dist, dev)
        winner = multi.best_restart(t_inertia, first_restart, dist, dev)   # the job's result: (restart id, inertia, owner)
        dist.barrier()
    else:
        winner = multi.best_restart(t_inertia.cpu(), first_restart)
    units_per_step = N * R * iters
    value = world * units_per_step * args.steps / elapsed

    # ---------------- multi-rank only: the real separator with the SCAFFOLDS sharded over the ranks -----------
    # (SURVEY §8(e), configs[4] shape: every rank holds its own scaffolds; per node one integer all-reduce of the
    # histograms + min/max of the SCG flip values.)  Weak scaling: rank k holds a table of the configs[1] size
    # generated from its own seed; the node timed is the root of the union.
    sep_sharded = None
    if dist is not None and not args.no_sep_sharded:
        keys_k, dp2scaf_k, scaf_len_k, _g, _G = make_points(20241 + 1000 * rank, args.scaffolds, args.samples, dev)
        keys_k_host = keys_k.cpu().numpy()
        t0 = time.perf_counter()
        dv, do = multi.global_dictionary(keys_k_host, dist, dev)
        dict_s = time.perf_counter() - t0
        sctx = _lib.Context(dev_index)
        sstore = separator.ShardedColumnStore(sctx, keys_k_host, dp2scaf_k, scaf_len_k, dv, do)
        nd = separator.separate_sharded(sstore, None, dist, dev)          # warm-up (RCCL channels, buffers)
        barrier()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            nd = separator.separate_sharded(sstore, None, dist, dev)
        torch.cuda.synchronize()
        sh_s = multi.max_elapsed((time.perf_counter() - t0) / reps, dist, dev)
        n_all = multi.sum_counts([keys_k.shape[1]], dist, dev)[0]
        sep_sharded = dict(value=D * n_all / sh_s, unit="dimension*points/s", ms_per_node=sh_s * 1e3, node="root of the union of the ranks' tables",
                           dps_total=int(n_all), split=int(nd.out.split), dimension=int(nd.out.dimension), key=int(nd.out.key),
                           exchange_bytes_per_node=int(nd.hist.numel() * 8), dictionary_s=round(dict_s, 2),
                           collective=f"all-reduce SUM int64 x{nd.hist.numel()} per node over {args.backend}")
        del sstore, nd
        sctx.close()
        dist.barrier()

    # ---------------- multi-rank only: k-means with the POINTS sharded over the ranks ---------------------------
    # (SURVEY §8(e) second way, configs[2] shape: every rank holds its own points, the same R restarts run over
    # the union; per pass one exchange of the cluster-1 partial sums [Rp][W] fp64, added in rank order.)
    km_sharded = None
    if dist is not None and not args.no_sep_sharded:
        n_loc = (N // 32) * 32                                              # whole MFMA tiles per shard
        x_loc = x_t[:, :n_loc].contiguous()
        n_tot = n_loc * world
        t_a = torch.empty((R, n_loc), dtype=torch.uint8, device=dev)
        t_c2 = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
        t_i2 = torch.empty(R, dtype=torch.float64, device=dev)
        it_sh = max(2, iters // 4)
        kmeans.kmeans_sharded_dev(ctx, x_loc, rank * n_loc, n_tot, R, 20240, 1, args.chunk, dist, t_a, t_c2, t_i2)   # warm-up
        barrier()
        t0 = time.perf_counter()
        kmeans.kmeans_sharded_dev(ctx, x_loc, rank * n_loc, n_tot, R, 20240, it_sh, args.chunk, dist, t_a, t_c2, t_i2)
        ctx.synchronize(); torch.cuda.synchronize()
        ks = multi.max_elapsed(time.perf_counter() - t0, dist, dev)
        chk = t_i2.clone()
        multi._all_reduce(chk, dist.ReduceOp.MAX, dist)                      # every rank must hold the same inertia bits
        km_sharded = dict(value=n_tot * R * it_sh / ks, unit="point*iterations/s", points_total=int(n_tot), restarts=R, iterations=it_sh,
                          ms_per_iteration=round(ks / (it_sh + 1) * 1e3, 3), identical_on_all_ranks=bool(torch.equal(chk, t_i2)),
                          exchange_bytes_per_pass=int(8 * ((R + 31) // 32 * 32) * ((D + 1 + 31) // 32 * 32)),
                          collective=f"all-gather + rank-ordered fp64 adds per pass over {args.backend}")
        del x_loc, t_a
        dist.barrier()

    # ---------------- multi-rank only: signature build, scaffolds dealt to the ranks (no exchange) ---------------
    sig_sharded = None
    if dist is not None and not args.no_sep_sharded:
        mbp_k = max(100, args.sig_mbp // 2)                                 # weak scaling: every rank its own assembly
        blob_k, off_k = make_assembly_blob(20244 + 7 * rank, mbp_k)
        tot_k = len(blob_k)
        ta = torch.from_numpy(np.concatenate([blob_k, np.zeros((-tot_k) % 16 + 16, np.uint8)])).to(dev)
        to = torch.from_numpy(off_k.astype(np.int64)).to(dev)
        cap_k = tot_k // 2000 + len(off_k) + 16
        ts_, tst, ten = (torch.empty(cap_k, dtype=torch.int32, device=dev), torch.empty(cap_k, dtype=torch.int64, device=dev),
                         torch.empty(cap_k, dtype=torch.int64, device=dev))
        td = torch.empty((cap_k, 180), dtype=torch.float64, device=dev)

        def sig_k():
            nk = signature.segment_dev(ctx, ta, tot_k, to, 2000, ts_, tst, ten)
            signature.count_dev(ctx, ta, tot_k, to, ts_, tst, ten, nk, t_dims=td)
            return nk
        n_k = sig_k()
        barrier()
        t0 = time.perf_counter()
        for _ in range(3):
            sig_k()
        ctx.synchronize()
        sk = multi.max_elapsed((time.perf_counter() - t0) / 3, dist, dev)
        segs = multi.sum_counts([n_k], dist, dev)[0]
        sig_sharded = dict(value=world * tot_k / sk / 1e9, unit="Gbp/s", mbp_per_rank=mbp_k, segments_total=int(segs),
                           ms_per_pass=round(sk * 1e3, 3), collective="none (rows concatenate in scaffold order)")
        del ta, td
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---------------- roofline of the dominant kernel --------------------------------------------
    # Flops per unit (point x restart x iteration).  SURVEY §8(d) prices the textbook Lloyd step at
    # d*(3k+1) = 7d.  For k = 2 the library runs one discriminant dot product (2(d+1)) and the masked
    # sums of ONE cluster (2(d+1)); the other cluster is totals minus that.  `achieved` is priced on
    # the 4(d+1) flops the MFMA pipe is actually asked for, so frac stays a utilisation (<= 1);
    # the 7d-based figure is reported beside it (it can exceed the peak: work the algebra removed).
    flops_per_unit = 4 * (D + 1)
    ms_avg = ms_launch / max(n_launch, 1)
    achieved = N * R * flops_per_unit / (ms_avg * 1e-3) / 1e12
    # HBM bytes per launch from the committed PMC passes (profiles/pmc_r01.json; collected on this
    # workload with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as the microarchitecture guide says)
    pmc_all = {}
    try:
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_r01.json")))
    except Exception:
        pmc_all = {}
    default_workload = (args.scaffolds, args.samples, R, args.sig_mbp) == (200000, 10, 64, 1000)
    traffic = pmc_all.get("km_step_kernel", {}).get("hbm_bytes_per_launch") if default_workload else None
    roofline = dict(kernel="km_step_kernel", bound="mfma", achieved=round(achieved, 3), peak=FP32_MFMA_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                    launches=n_launch, ms_per_launch=round(ms_avg, 4), flops_per_unit=flops_per_unit,
                    survey_flops_per_unit=7 * D,
                    achieved_at_survey_count=round(N * R * 7.0 * D / (ms_avg * 1e-3) / 1e12, 3),
                    hbm_algorithmic_GBs=round((4.0 * N * D + 4.0 * R * (D + 1)) / (ms_avg * 1e-3) / 1e9, 1))

    out = dict(metric="k-means point·iterations/sec (whole node) + signature build Gbp/s",
               value=value, unit="point*iterations/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak", vs_baseline=None,
               dtype="f32", data="synthetic",
               config=dict(workload=f"configs[1]: {args.scaffolds} scaffolds -> {N} dps, 179 signature + {args.samples} coverage dims "
                                    f"(z-scored), k=2, {R} restarts x {iters} Lloyd iterations per GPU, chunk {args.chunk}",
                           parallelism=f"restarts x{world}" if world > 1 else "single GPU"),
               roofline=roofline, best_restart=dict(id=int(winner[0]), inertia=float(winner[1]), rank=int(winner[2])))
    if sep_sharded is not None:
        out["separator_sharded"] = sep_sharded
    if km_sharded is not None:
        out["kmeans_point_sharded"] = km_sharded
    if sig_sharded is not None:
        out["signature_sharded"] = sig_sharded

    # ---------------- extras: real separator + signature build on this GPU -----------------------
    keys_host = None
    if not args.no_extras:
        keys_host = keys_t.cpu().numpy()
        t0 = time.perf_counter()
        store = separator.ColumnStore(ctx, keys_host, dp2scaf, scaf_len)
        load_s = time.perf_counter() - t0
        sep = separator.ClusterSeparatorHip(store)
        sep.separate()
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        reps = 3
        call_s = 0.0
        for _ in range(reps):
            sep.separate()
            call_s += sep.last_call_s
        wrap_s = (time.perf_counter() - t0) / reps            # with the Python glue (numpy buffers, ctypes structs)
        sep_s = call_s / reps                                 # abw_sep_node alone: host node list in, host sides out
        na, msa = ctx.prof_read("sep_accum")
        ne, mse = ctx.prof_read("sep_eval")
        ctx.prof_enable(False)
        alg_bytes = 4.0 * D * N + 4.0 * N + 16.0 * D * 1001
        out["separator"] = dict(value=D * N / sep_s, unit="dimension*points/s", ms_per_node=sep_s * 1e3, ms_per_node_with_python_glue=round(wrap_s * 1e3, 3), node="root",
                                split=int(sep._res.split), dimension=int(sep._res.dimension), load_s=round(load_s, 2),
                                roofline=dict(kernel="sep_accum_kernel", bound="hbm", achieved=round(alg_bytes / (msa / na * 1e-3) / 1e9, 1),
                                              peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg_bytes / (msa / na * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              traffic=pmc_all.get("sep_accum_kernel", {}).get("hbm_bytes_per_launch_raw") if default_workload else None,
                                              ms_per_launch=round(msa / na, 4), eval_ms=round(mse / max(ne, 1), 4)))
        # the whole recursion on this table (reference src/abawaca.cpp:95-197 with the GPU separator): every node is one
        # abw_sep_node call on the resident column store, given its scaffold list
        t0 = time.perf_counter()
        nodes_all, s2c_all, _d2c = separator.binning(store)
        rec_s = time.perf_counter() - t0
        bins = [r for r in nodes_all if not r["split"]]
        # planted genomes vs bins: share of scaffolds whose bin's majority genome is their own
        lab = np.asarray(genome)
        agree = 0
        for r in bins:
            mem = lab[s2c_all == r["id"]]
            if len(mem):
                agree += int(np.bincount(mem).max())
        out["separator"]["recursion"] = dict(s=round(rec_s, 3), c_abi_s=round(sum(r["call_s"] for r in nodes_all), 3), nodes=len(nodes_all), bins=len(bins), ms_per_node=round(rec_s / len(nodes_all) * 1e3, 3),
                                             planted_genomes=int(G), scaffolds_in_their_bins_majority_genome=round(agree / len(lab), 4),
                                             note="whole recursive binning of the configs[1] table, GPU separator + Python work-list loop")
        del store

        blob, off = make_assembly_blob(20244, args.sig_mbp)
        total = len(blob)
        pad = (-total) % 16 + 16
        t_ascii = torch.from_numpy(np.concatenate([blob, np.zeros(pad, np.uint8)])).to(dev)
        t_off = torch.from_numpy(off.astype(np.int64)).to(dev)
        cap = total // 2000 + len(off) + 16
        t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
        t_st = torch.empty(cap, dtype=torch.int64, device=dev)
        t_en = torch.empty(cap, dtype=torch.int64, device=dev)
        t_dims = torch.empty((cap, 180), dtype=torch.float64, device=dev)
        torch.cuda.synchronize()

        def sig_step():
            n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
            signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims)
            return n
        n_seg = sig_step()
        ctx.synchronize()
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            sig_step()
        ctx.synchronize()
        sig_s = (time.perf_counter() - t0) / reps
        nc, msc = ctx.prof_read("sig_count")
        ns, mss = ctx.prof_read("sig_segment")
        ctx.prof_enable(False)
        alg = total + 1440.0 * n_seg
        out["signature"] = dict(value=total / sig_s / 1e9, unit="Gbp/s", mbp=args.sig_mbp, segments=int(n_seg),
                                ms_per_pass=sig_s * 1e3, resident="ASCII in HBM; segmentation + counting + fp64 finalisation timed",
                                roofline=dict(kernel="sig_count_kernel", bound="hbm", achieved=round(alg / (msc / nc * 1e-3) / 1e9, 1),
                                              peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg / (msc / nc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              traffic=pmc_all.get("sig_count_kernel", {}).get("hbm_bytes_per_launch") if default_workload else None,
                                              ms_per_launch=round(msc / nc, 4), segment_ms=round(mss / max(ns, 1), 4)))
        # the boundary with HOST buffers (abw_sig_build: H2D of the ASCII, segmentation + counting, D2H of the rows):
        # PCIe-inclusive, never the headline
        sub = min(total, 200_000_000)
        n_sub = int(np.searchsorted(off, sub, side="right")) - 1
        hb, ho = blob[:int(off[n_sub])], off[:n_sub + 1]
        signature.build_signatures(ctx, hb[:10_000_000], off[:1001], want_dims=False)      # warm-up (buffers)
        t0 = time.perf_counter()
        r_host = signature.build_signatures(ctx, hb, ho, want_dims=False)
        host_s = time.perf_counter() - t0
        out["signature"]["host_buffers"] = dict(value=round(len(hb) / host_s / 1e9, 2), unit="Gbp/s", mbp=len(hb) // 1_000_000, segments=int(len(r_host["seg_scaf"])),
                                                note="abw_sig_build on pageable host memory, milli rows back (716 B per segment): PCIe both ways + allocation of the outputs")
        del r_host
        # ---- row N1: coverage accumulation on the same segments (20 M reads of 150 bp ~ depth 3 of this assembly)
        n_scaf_sig = len(off) - 1
        n_reads = args.cov_mreads * 1_000_000
        gcov = torch.Generator(device=dev); gcov.manual_seed(20247)
        r_scaf = torch.randint(0, n_scaf_sig, (n_reads,), device=dev, generator=gcov, dtype=torch.int64)
        r_pos = (torch.rand(n_reads, device=dev, generator=gcov) * (10_000 - 150)).to(torch.int64)
        # SAM files come sorted by reference or not at all; keep the unsorted (worst) case
        t_rscaf = r_scaf.to(torch.int32); t_rpos = r_pos.contiguous(); t_rlen = torch.full((n_reads,), 150, dtype=torch.int32, device=dev)
        t_cov = torch.zeros(cap, dtype=torch.float64, device=dev)
        pv = lambda t: C.c_void_p(t.data_ptr())
        def cov_step():
            t_cov.zero_()
            torch.cuda.synchronize()
            ctx.check(_lib.lib().abw_cov_accumulate_dev(ctx.handle, pv(t_scaf), pv(t_st), pv(t_en), int(n_seg), n_scaf_sig,
                                                        pv(t_rscaf), pv(t_rpos), pv(t_rlen), n_reads, pv(t_cov)))
            ctx.synchronize()
        cov_step()
        ctx.prof_reset(); ctx.prof_enable(True)
        t0 = time.perf_counter()
        for _ in range(reps):
            cov_step()
        cov_s = (time.perf_counter() - t0) / reps
        ncv, mscv = ctx.prof_read("cov_accumulate")
        ctx.prof_enable(False)
        alg_cov = 16.0 * n_reads + 8.0 * 2 * n_seg
        out["coverage"] = dict(value=n_reads / cov_s / 1e6, unit="M reads/s", reads=n_reads, segments=int(n_seg), ms_per_sample=cov_s * 1e3,
                               note="row N1 (next): per-segment overlap sums in SAM order, fp64; reads and segments resident; "
                                    "includes a stable radix sort (rocPRIM) that brings each segment's contributions together in read order",
                               roofline=dict(kernel="cov_accumulate (4 kernels + sort)", bound="hbm", achieved=round(alg_cov / (mscv / ncv * 1e-3) / 1e9, 1),
                                             peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg_cov / (mscv / ncv * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                             traffic=None, ms_per_launch=round(mscv / ncv, 4)))
        if world == 1 and not args.no_cpu:
            import _oracle as orc
            ns = min(n_reads, 4_000_000)
            hs, hp, hl = t_rscaf[:ns].cpu().numpy().astype(np.uint32), t_rpos[:ns].cpu().numpy().astype(np.uint64), t_rlen[:ns].cpu().numpy().astype(np.uint32)
            ss, s0, s1 = t_scaf[:n_seg].cpu().numpy().astype(np.uint32), t_st[:n_seg].cpu().numpy().astype(np.uint64), t_en[:n_seg].cpu().numpy().astype(np.uint64)
            t0 = time.perf_counter()
            orc.coverage(ss, s0, s1, n_scaf_sig, hs, hp, hl)
            dt = time.perf_counter() - t0
            out["coverage"]["cpu_baseline"] = dict(value=ns / dt / 1e6, unit="M reads/s", cores=1, kind="port",
                                                   sample=f"oracle abw_or_coverage (the reference's loops, its SAM text parsing not included), "
                                                          f"first {ns} reads, {dt:.1f} s")
        del t_ascii, t_dims
        out["lrn"] = lrn_bench(ctx, world == 1 and not args.no_cpu)

    # ---------------- optional: configs[0], the reference's own CPU-runnable case, end to end -----------------
    if args.c1 and world == 1:
        out["recursion_c1"] = recursion_c1(ctx)

    # ---------------- CPU baselines (rank 0, N = 1 only) -------------------------------------------
    if world == 1 and not args.no_cpu:
        x_host = x_t[:, :min(N, 200_000)].cpu().numpy()
        out["cpu_baseline"] = cpu_kmeans_baseline(x_host, R, iters, args.chunk)
        out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        if not args.no_extras:
            b = cpu_signature_baseline(blob, off)
            if b:
                out["signature"]["cpu_baseline"] = b
            b = cpu_separator_baseline(keys_host, dp2scaf, scaf_len)
            if b:
                out["separator"]["cpu_baseline"] = b
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
