#!/usr/bin/env python3
"""bench.py — throughput of the abawaca hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
     -- or bare: with no WORLD_SIZE in the environment the process starts its own N ranks, one fresh child per GPU, before it
     touches a GPU, and exits with their status; n_gpus in the line is always the number of ranks that ran.)

Headline (BASELINE.json "metric"): k-means point·iterations/s over the whole node, plus the signature
build in Gbp/s and the real separator in dimension·points/s as extra keys of the same JSON line.  A
"step" is one full pass: one k-means run of `--iters` Lloyd iterations over all restarts, inputs already
resident in HBM.

  N = 1 : configs[1] (200k scaffolds -> ~0.8 M dps, 179 signature + 10 coverage dims, 64 restarts) -- the
          configuration the metric is quoted on; configs[2]'s job on the one GPU is an extra key.
  N > 1 : configs[2] -- ONE job of 256 restarts on the 1M-scaffold table (~4.0 M dps x 199 dims), strong
          scaling, under both partitions the north-star names:
            "restarts" : rank k runs restarts [256k/N, 256(k+1)/N) on a replica of the points; no data-path
                         collective, one gather of the inertias picks the winner
            "points"   : rank k holds 1/N of the points and runs all 256 restarts; per pass one exchange of
                         the cluster-1 partial sums (all-gather + rank-ordered fp64 adds over RCCL)
          value = the better of the two (both are in the line); the old weak-scaling run (64 restarts per
          GPU on the configs[1] table) stays as the extra key "weak_restarts".  The scaffold-sharded
          separator leg is configs[4]'s shape: 2M scaffolds in total, 1/N of them on every rank.

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
    ap.add_argument("--scaffolds", type=int, default=0, help="0 = the configuration's own (configs[1]: 200 000; N > 1, configs[2]: 1 000 000)")
    ap.add_argument("--samples", type=int, default=0, help="0 = the configuration's own (10 / 20)")
    ap.add_argument("--restarts", type=int, default=0, help="restarts of the whole job; 0 = the configuration's own (64 / 256)")
    ap.add_argument("--genomes", type=int, default=0, help="planted genomes; 0 = the configuration's own (100 / 200)")
    ap.add_argument("--sep-scaffolds", type=int, default=2_000_000, help="N > 1: scaffolds of the sharded separator leg in total (configs[4])")
    ap.add_argument("--sep-nodes", type=int, default=127, help="N > 1, with --sep-node-by-node: nodes of that recursion to run (0 = all)")
    ap.add_argument("--sep-node-by-node", action="store_true", help="N > 1: also the older scaffold-sharded separator that exchanges every histogram "
                    "bin of the table per NODE (abw_sep_shard_*), as separator_sharded_node_by_node")
    ap.add_argument("--sep-nodes-dims", type=int, default=0, help="N > 1: nodes of the dimension-sharded recursion to run (0 = all)")
    ap.add_argument("--no-split", action="store_true", help="N = 1: skip the same job in the split-bf16 arithmetic (abw_km_set_arith)")
    ap.add_argument("--no-c2", action="store_true", help="N = 1: skip the configs[2] job on the one GPU")
    ap.add_argument("--no-c4", action="store_true", help="N = 1: skip configs[4]'s recursion (2M scaffolds) on the one GPU")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--chunk", type=int, default=0, help="points per fp32 partial sum; 0 = the library's choice (abw_km_auto_chunk)")
    ap.add_argument("--sig-mbp", type=int, default=1000, help="signature side bench: assembly size in Mbp")
    ap.add_argument("--sig10-mbp", type=int, default=10_000, help="configs[3] leg: assembly size in Mbp (0 = skip)")
    ap.add_argument("--cov-mreads", type=int, default=20, help="coverage side bench: million reads of 150 bp")
    ap.add_argument("--no-extras", action="store_true", help="only the k-means headline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1: nccl (= RCCL over xGMI); "
                    "gloo lets several ranks share one GPU for rehearsals (collectives staged through the host)")
    ap.add_argument("--no-sep-sharded", action="store_true", help="skip the scaffold-sharded separator leg of a multi-rank run")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baselines")
    ap.add_argument("--c1", action="store_true", help="also run configs[0] end to end: the whole recursive binning of a 10k-scaffold "
                    "table, reference driver (its own separator, all host cores; takes about a minute) beside the GPU separator")
    return ap.parse_args()


def make_points(seed, n_scaf, n_samples, device, n_genomes=0):
    """configs[1]-shaped feature table generated on the GPU (same model as abawaca_amd.synth.make_features:
    per-genome GC -> expected canonical k-mer frequencies + sampling noise, truncated to thousandths;
    coverage c_{g,s}(1+N(0,0.1))).  -> (keys int32 [D, N] on device, dp2scaf, scaf_len, genome)"""
    import torch
    from abawaca_amd import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    G = n_genomes or max(2, n_scaf // 2000)
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


def build_dropin_leg(blob, off, mbp_sample=200):
    """The compiled drop-in for `abawaca-build` end to end (oracle/_ref/hip_build_driver = integration/abawaca_build_hip.cpp linked with the
    reference's own FASTA / SAM readers) beside the reference's program on the same FASTA + one SAM file: wall clock of both,
    where the drop-in's time goes, and whether the three output files are the same bytes."""
    ref_exe = os.path.join(ROOT, "oracle", "_ref", "abawaca-build")
    hip_exe = os.path.join(ROOT, "oracle", "_ref", "hip_build_driver")
    if not (os.path.exists(ref_exe) and os.path.exists(hip_exe)):
        return None
    n_scaf = min(len(off) - 1, mbp_sample * 100)
    end = int(off[n_scaf])
    with tempfile.TemporaryDirectory(prefix="abwbuild_") as wd:
        fa = os.path.join(wd, "a.fa")
        with open(fa, "wb") as f:
            for i in range(n_scaf):
                f.write(b">s%d\n" % i)
                f.write(blob[int(off[i]):int(off[i + 1])].tobytes())
                f.write(b"\n")
        # one sample whose reads reach every segment (the reference prints a segment without reads from memory it never wrote)
        sam = os.path.join(wd, "s0.sam")
        tail = b"\t255\t150M\t*\t0\t0\t" + b"A" * 150 + b"\t" + b"I" * 150 + b"\n"
        n_reads = 0
        with open(sam, "wb") as f:
            f.write(b"@HD\tVN:1.0\n")
            for i in range(n_scaf):
                L = int(off[i + 1] - off[i])
                f.write(b"".join(b"r%d\t0\ts%d\t%d" % (n_reads + k, i, pos) + tail for k, pos in enumerate(range(1, max(2, L - 150), 700))))
                n_reads += len(range(1, max(2, L - 150), 700))
        res = {}
        outs = {}
        for tag, exe in (("reference", ref_exe), ("dropin", hip_exe)):
            out = os.path.join(wd, "out_" + tag)
            os.makedirs(out)
            times = os.path.join(wd, tag + ".times.json")
            t0 = time.perf_counter()
            subprocess.run([exe, "-f", fa, "-o", out, "-s", sam, "-c", sam], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           env=dict(os.environ, ABW_BUILD_TIMES=times))
            res[tag] = dict(wall_s=round(time.perf_counter() - t0, 2))
            if os.path.exists(times):
                res[tag]["stages"] = json.load(open(times))
            outs[tag] = [open(os.path.join(out, "abawaca." + k), "rb").read() for k in ("lrn", "names", "info")]
    d, r = res["dropin"], res["reference"]
    return dict(value=end / d["wall_s"] / 1e9, unit="Gbp/s (whole program, wall clock: process start, FASTA + SAM read, GPU, files written)",
                mbp=round(end / 1e6), reads=n_reads, wall_s=d["wall_s"], stages=d.get("stages"),
                identical_files=bool(outs["dropin"] == outs["reference"]), speedup_over_reference_program=round(r["wall_s"] / d["wall_s"], 1),
                note="the drop-in's wall clock is the reference's own single-threaded FASTA reader (ref src/SeqIORead_fasta.h:51-103) and "
                     "process start-up (HIP context); sig_build_on_the_worker_s, which overlaps the reader, is all the GPU stage costs",
                cpu_baseline=dict(value=end / r["wall_s"] / 1e9, unit="Gbp/s", cores=1, kind="reference",
                                  sample=f"reference abawaca-build (-O2 build of its own sources) on the same {end / 1e6:.0f} Mbp + {n_reads} reads, "
                                         f"whole program, {r['wall_s']:.1f} s"))


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


def side_leg(name, fn, *a, **kw):
    """A side leg must not cost the run its headline line: what it raises is reported in its place."""
    try:
        return fn(*a, **kw)
    except Exception as e:                                   # noqa: BLE001 -- reported, not hidden
        import traceback
        traceback.print_exc(file=sys.stderr)
        return dict(error=f"{name}: {type(e).__name__}: {e}"[:500])


def sam_leg(ctx, dev, with_cpu, n_blocks=100, block_lines=20000, n_names=100_000):
    """Row N1, the reader side: SAM text resident in HBM -> the arrays abw_cov_accumulate takes (abw_sam_parse_dev)."""
    import subprocess
    import tempfile
    import torch
    from _sam_gen import random_sam
    from abawaca_amd import _lib, coverage
    names = ["scaffold_%d" % i for i in range(n_names)]
    block = random_sam(20248, block_lines, names)
    head = int(np.flatnonzero(block == 10)[5]) + 1                     # the six header lines once
    body = block[head:]
    text = np.concatenate([block[:head], np.tile(body, n_blocks)])
    n_lines = block_lines * n_blocks
    t_text = torch.from_numpy(text).to(dev)
    blob, off = coverage.names_table(names)
    cap = n_lines + 8
    t_s, t_p, t_l = (torch.empty(cap, dtype=torch.int32, device=dev), torch.empty(cap, dtype=torch.int64, device=dev),
                     torch.empty(cap, dtype=torch.int32, device=dev))
    cnt = coverage.SamCounts()
    pv = lambda t: C.c_void_p(t.data_ptr())
    pn = lambda a: C.c_void_p(a.ctypes.data)

    def step():
        ctx.check(_lib.lib().abw_sam_parse_dev(ctx.handle, pv(t_text), len(text), pn(blob), pn(off), len(names), 15, cap, pv(t_s), pv(t_p), pv(t_l),
                                               None, None, None, None, None, C.byref(cnt)))
    step()
    ctx.synchronize()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        step()
    ctx.synchronize()
    wall = (time.perf_counter() - t0) / reps
    nk, msk = ctx.prof_read("sam_parse")
    ctx.prof_enable(False)
    ms = msk / nk
    alg = float(len(text)) + 16.0 * int(cnt.kept)
    res = dict(value=len(text) / (ms * 1e-3) / 1e9, unit="GB/s of SAM text (kernels, text resident)", records=int(cnt.records), kept=int(cnt.kept),
               text_mb=round(len(text) / 1e6, 1), ms_kernels=round(ms, 3), ms_call=round(wall * 1e3, 3), records_per_s=round(int(cnt.records) / (ms * 1e-3) / 1e6, 1),
               note="row N1 (next), the reader side; records_per_s in M records/s; ms_call includes building and uploading the hash table of the "
                    f"{n_names} scaffold names; the text is {n_blocks} x the same {block_lines} generated records (every kind the reference's reader tells apart)",
               roofline=dict(kernel="sam_parse_kernel (+ line index, compaction)", bound="hbm", achieved=round(alg / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                             unit="GB/s", frac=round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=None))
    exe = os.path.join(ROOT, "oracle", "_ref", "ref_sam_reader")
    if with_cpu and os.path.exists(exe):
        sample = np.concatenate([block[:head], np.tile(body, n_blocks)])
        with tempfile.TemporaryDirectory(prefix="abwsam_") as wd:
            p = os.path.join(wd, "s.sam")
            sample.tofile(p)
            o = subprocess.run([exe, p], capture_output=True, text=True, timeout=600).stdout.split()
        if len(o) >= 7:
            # the same sample through the library: the reference's own reader must count the same records
            c2 = coverage.SamCounts()
            t_smp = torch.from_numpy(sample).to(dev)
            ctx.check(_lib.lib().abw_sam_parse_dev(ctx.handle, pv(t_smp), len(sample), pn(blob), pn(off), len(names), 15, cap, pv(t_s), pv(t_p), pv(t_l),
                                                   None, None, None, None, None, C.byref(c2)))
            res["cpu_baseline"] = dict(value=round(len(sample) / float(o[6]) / 1e9, 4), unit="GB/s of SAM text", cores=1, kind="reference",
                                       sample=f"the reference's SAMReader + ReadMapping (oracle/_ref/ref_sam_reader, -O2, single-threaded by design) on "
                                              f"{int(o[0])} of the records, {float(o[6]):.1f} s",
                                       same_counts=bool(int(o[0]) == int(c2.records) and int(o[1]) == int(c2.kept)))
    return res


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
               roofline=dict(kernel="lrn_blocks_kernel (+ line scan)", bound="hbm", achieved=round(alg / k_s / 1e9, 1), peak=HBM_PEAK_GBS,
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
                         roofline=dict(kernel="lrn_emit3_kernel", bound="hbm", achieved=round((4.0 * D * N + len(out_text)) / e_s / 1e9, 1),
                                       peak=HBM_PEAK_GBS, unit="GB/s", frac=round((4.0 * D * N + len(out_text)) / e_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                                       note="algorithmic bytes: every 4-byte value once + the text once (one launch since round 3; "
                                            "rounds 1-2 read the values twice and were quoted on 8 bytes per value)"))
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
    alg_c = 8.0 * D * N + cbytes                              # every value read once, the text written once (one launch since round 3)
    res["cluster_writer"] = dict(value=cbytes / c_s / 1e9, unit="GB/s of text (kernels)", clusters=n_cl, rows_per_cluster=per, text_mb=round(cbytes / 1e6, 1),
                                 ms_kernels=round(c_s * 1e3, 3), ms_wall_text_left_on_device=round(wall_c * 1e3, 2),
                                 roofline=dict(kernel="lrn_emit6_kernel", bound="hbm", achieved=round(alg_c / c_s / 1e9, 1),
                                               peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg_c / c_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                                               note="algorithmic bytes: every value once + the text once; one call per cluster of 2 000 rows is launch-bound"))
    # the same clusters in ONE call (concatenated lists, row offsets back)
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
                                             roofline_frac=round(alg_c / o_s / 1e9 / HBM_PEAK_GBS, 4),
                                             note="the caller's [columns][datapoints] table: a 128-byte line per value read")
    # from the resident copy (abw_lrn_table_load_dev keeps the table transposed: a row is one coalesced read) -- what
    # integration/ClusterEsomWriterHip prints from
    shape = lrn.lrn_table_load_dev(ctx, t_vals)
    def clusters_resident():
        tot = 0
        for c in range(n_cl):
            tot += lrn.lrn_emit_rows_dev(ctx, None, perm[c * per:(c + 1) * per], t_out, resident=shape)
        return tot
    clusters_resident()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    rbytes = clusters_resident()
    wall_r = time.perf_counter() - t0
    nr, msr = ctx.prof_read("lrn_emit_rows")
    t_chk = torch.empty(cbytes, dtype=torch.uint8, device="cuda")
    lrn.lrn_emit_rows_dev(ctx, None, perm[:n_cl * per], t_chk, t_off, resident=shape)
    ctx.prof_reset()
    for _ in range(reps):
        lrn.lrn_emit_rows_dev(ctx, None, perm[:n_cl * per], t_chk, t_off, resident=shape)
    nro, msro = ctx.prof_read("lrn_emit_rows")
    ctx.prof_enable(False)
    assert rbytes == cbytes and torch.equal(t_chk, t_all)
    r_s, ro_s = msr * 1e-3, msro / reps * 1e-3
    res["cluster_writer"]["resident_copy"] = dict(
        value=cbytes / r_s / 1e9, unit="GB/s of text (kernels)", ms_kernels=round(r_s * 1e3, 3), ms_wall_text_left_on_device=round(wall_r * 1e3, 2),
        roofline_frac=round(alg_c / r_s / 1e9 / HBM_PEAK_GBS, 4),
        one_call=dict(value=cbytes / ro_s / 1e9, unit="GB/s of text (kernels)", ms_kernels=round(ro_s * 1e3, 3),
                      roofline_frac=round(alg_c / ro_s / 1e9 / HBM_PEAK_GBS, 4)))
    lrn.lrn_table_load_dev(ctx, None)
    del t_chk
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


def make_scgs(seed, genome, G, n_scg=30):
    """30 SCG names, 30 carrying scaffolds per genome.  -> (scg_scaf int32[M], scg_id int32[M])"""
    rng = np.random.Generator(np.random.PCG64(seed + 77))
    order = np.argsort(genome, kind="stable")
    bounds = np.searchsorted(genome[order], np.arange(G + 1))
    scg_scaf, scg_id = [], []
    for g in range(G):
        members = order[bounds[g]:bounds[g + 1]]
        if len(members) == 0:
            continue
        k = min(n_scg, len(members))
        scg_scaf.append(rng.choice(members, size=k, replace=False))
        scg_id.append(np.arange(k))
    return np.concatenate(scg_scaf).astype(np.int32), np.concatenate(scg_id).astype(np.int32)


# BASELINE.json configs (SURVEY §8(d) generator parameters)
CONFIGS = {1: dict(scaffolds=200_000, samples=10, genomes=100, restarts=64),
           2: dict(scaffolds=1_000_000, samples=20, genomes=200, restarts=256),
           4: dict(scaffolds=2_000_000, samples=20, genomes=400)}


PROF_PERIOD = 7     # the timed region brackets every 7th launch of a kernel family with HIP events (three of a job's 21 passes): a
                    # pair costs the stream ~9 us (experiments/km_prof_cost.py: 7.9 -> 8.3 ms per configs[1] job with every launch
                    # bracketed); the launches of a family all do the same work


def timed_steps(step, warmup, steps, barrier, ctx, prof_name="km_step"):
    """The bench contract: `warmup` untimed steps, then EXACTLY `steps` steps between barrier + synchronize on both sides.
    -> (elapsed seconds of this rank, launches of `prof_name` that were timed -- every PROF_PERIOD-th --, their summed ms)"""
    import torch
    for _ in range(warmup):
        step()
    barrier()
    ctx.prof_reset()
    ctx.prof_enable(PROF_PERIOD)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    n_launch, ms_launch = ctx.prof_read(prof_name)
    ctx.prof_enable(False)
    return elapsed, n_launch, ms_launch


def host_build_leg(ctx, blob, off, window=2000):
    """abw_sig_build with HOST buffers in and out (the boundary the reference's FASTA loop would call): pageable input,
    pre-faulted pageable outputs (a caller's re-used buffers), milli rows (716 B per segment) back."""
    from abawaca_amd import _lib
    L = _lib.lib()
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    offu = np.ascontiguousarray(off, dtype=np.uint64)
    n_scaf = len(offu) - 1
    cap = int(len(blob) // window + n_scaf + 16)
    seg_scaf = np.zeros(cap, dtype=np.uint32); st = np.zeros(cap, dtype=np.uint64); en = np.zeros(cap, dtype=np.uint64)
    milli = np.zeros((cap, 179), dtype=np.int32)
    n = C.c_uint64(0)
    p = lambda a: C.c_void_p(a.ctypes.data)

    def run(nsc):
        ctx.check(L.abw_sig_build(ctx.handle, p(blob), p(offu), nsc, window, cap, p(seg_scaf), p(st), p(en), None, p(milli), C.byref(n)))
    run(min(n_scaf, 20_000))                                  # warm-up: streams, pinned ring, worker threads
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        run(n_scaf)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    k = int(n.value)
    bytes_in, bytes_out = len(blob), k * (4 + 8 + 8 + 179 * 4)
    return dict(value=round(len(blob) / best / 1e9, 2), unit="Gbp/s", mbp=len(blob) // 1_000_000, segments=k, s=round(best, 4),
                pcie=dict(h2d_GBs=round(bytes_in / best / 1e9, 1), d2h_GBs=round(bytes_out / best / 1e9, 1),
                          bound="PCIe 5.0 x16: ~55-60 GB/s each way in practice, i.e. <= ~55 Gbp/s of ASCII whatever the kernels do"),
                note="abw_sig_build on pageable host memory: chunks of whole scaffolds staged through pinned buffers by worker threads, "
                     "H2D | segmentation + counting | D2H overlapped; milli rows back (716 B per segment) into pre-faulted arrays; best of 3")


def make_assembly_blob_gpu(seed, mbp, dev):
    """make_assembly_blob's shape generated on the GPU (10 kbp scaffolds, iid ACGT, a run of 50 N in 2 % of them):
    -> (uint8 tensor on `dev`, padded for 16-byte reads; total bases; int64 offsets on `dev`)"""
    import torch
    n_scaf, Ls = mbp * 100, 10_000
    total = n_scaf * Ls
    g = torch.Generator(device=dev); g.manual_seed(seed)
    t = torch.empty(total + 32, dtype=torch.uint8, device=dev)
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    step = 1 << 28
    for o in range(0, total, step):
        m = min(step, total - o)
        t[o:o + m] = lut[torch.randint(0, 4, (m,), device=dev, generator=g, dtype=torch.uint8).to(torch.int64)]
    t[total:] = 0
    withn = torch.randperm(n_scaf, device=dev, generator=g)[: max(1, n_scaf // 50)]
    at = withn * Ls + torch.randint(0, Ls - 50, (len(withn),), device=dev, generator=g)
    idx = (at[:, None] + torch.arange(50, device=dev)[None, :]).reshape(-1)
    t[idx] = ord("N")
    off = torch.arange(n_scaf + 1, dtype=torch.int64, device=dev) * Ls
    return t, total, off


def signature_10g_leg(ctx, dev, mbp):
    """configs[3]: the signature build of a 10 Gbp assembly (1M scaffolds of 10 kbp), resident in HBM (10 GB of ASCII in,
    7.2 GB of fp64 rows out) and streamed from host memory through abw_sig_build."""
    import torch
    from abawaca_amd import signature
    t_ascii, total, t_off = make_assembly_blob_gpu(20243, mbp, dev)
    cap = total // 2000 + t_off.numel() + 16
    t_scaf = torch.empty(cap, dtype=torch.int32, device=dev)
    t_st = torch.empty(cap, dtype=torch.int64, device=dev)
    t_en = torch.empty(cap, dtype=torch.int64, device=dev)
    t_dims = torch.empty((cap, 180), dtype=torch.float64, device=dev)
    torch.cuda.synchronize()

    def step():
        n = signature.segment_dev(ctx, t_ascii, total, t_off, 2000, t_scaf, t_st, t_en)
        signature.count_dev(ctx, t_ascii, total, t_off, t_scaf, t_st, t_en, n, t_dims=t_dims)
        return n
    n_seg = step()
    ctx.synchronize()
    ctx.prof_reset(); ctx.prof_enable(True)
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        step()
    ctx.synchronize()
    dt = (time.perf_counter() - t0) / reps
    nc, msc = ctx.prof_read("sig_count")
    ns, mss = ctx.prof_read("sig_segment")
    ctx.prof_enable(False)
    alg = total + 1440.0 * n_seg
    # every segment's 180 dimensions sum to 4 (one per k): a checksum of the whole table
    chk = float(t_dims[:n_seg].sum()) / n_seg
    res = dict(workload=f"configs[3]: {mbp} Mbp, {t_off.numel() - 1} scaffolds of 10 kbp", value=total / dt / 1e9, unit="Gbp/s", segments=int(n_seg),
               ms_per_pass=dt * 1e3, resident="ASCII in HBM; segmentation + counting + fp64 finalisation timed", mean_row_sum=round(chk, 9),
               roofline=dict(kernel="sig_count_kernel", bound="hbm", achieved=round(alg / (msc / nc * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
                             frac=round(alg / (msc / nc * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), traffic=None, ms_per_launch=round(msc / nc, 4),
                             segment_ms=round(mss / max(ns, 1), 4)))
    del t_dims, t_scaf, t_st, t_en
    blob = t_ascii[:total].cpu().numpy()
    off = t_off.cpu().numpy()
    del t_ascii
    torch.cuda.empty_cache()
    res["host_buffers"] = host_build_leg(ctx, blob, off)
    return res


def separator_leg(ctx, keys_host, dp2scaf, scaf_len, genome, G, seed, label, traffic=None):
    """The real separator (reference ClusterSeparatorBySensitivitySpecificity with the SCG gate) on a resident table: the
    root node through abw_sep_node, then the whole recursion of reference src/abawaca.cpp:95-197."""
    from abawaca_amd import separator
    D, N = keys_host.shape
    S = len(scaf_len)
    scg_scaf, scg_id = make_scgs(seed, genome, G)
    so, si = separator.scg_csr(scg_scaf, scg_id, S)
    # the table from keys that are on the device already (where a GPU loader leaves them), then from host memory
    import torch
    t_keys = torch.from_numpy(keys_host).to(f"cuda:{torch.cuda.current_device()}")
    torch.cuda.synchronize()
    load_dev_s = None
    for _ in range(2):                                    # the second time: buffers of the context are grown
        t0 = time.perf_counter()
        store = separator.ColumnStore(ctx, None, dp2scaf, scaf_len, so, si, 30, device_keys=(t_keys.data_ptr(), D, N))
        load_dev_s = time.perf_counter() - t0
        del store
    del t_keys
    t0 = time.perf_counter()
    store = separator.ColumnStore(ctx, keys_host, dp2scaf, scaf_len, so, si, 30)
    load_s = time.perf_counter() - t0
    sep = separator.ClusterSeparatorHip(store, per_dim=False)       # the node's winner, as the reference's separate() reports it
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
    nw, msw = ctx.prof_read("sep_wide")
    ctx.prof_enable(False)
    # every pass over the table counts: the accumulation bracket (narrow histograms, SCG lists, the wide dimensions' tile sums)
    # and the wide dimensions' second pass (evaluation along the sorted positions)
    msa, accum_only_ms = msa + msw, msa / na
    alg_bytes = 4.0 * D * N + 4.0 * N + 16.0 * D * 1001
    res = dict(value=D * N / sep_s, unit="dimension*points/s", ms_per_node=sep_s * 1e3, ms_per_node_with_python_glue=round(wrap_s * 1e3, 3), node="root",
               table=f"{label}: {S} scaffolds -> {N} dps x {D} dims, 30 SCGs x {G} genomes",
               split=int(sep._res.split), dimension=int(sep._res.dimension), load_s=round(load_s, 3), load_from_device_keys_s=round(load_dev_s, 3),
               roofline=dict(kernel="sep_accum_kernel", bound="hbm", achieved=round(alg_bytes / (msa / na * 1e-3) / 1e9, 1),
                             peak=HBM_PEAK_GBS, unit="GB/s", frac=round(alg_bytes / (msa / na * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                             traffic=traffic, ms_per_launch=round(msa / na, 4), accumulate_ms=round(accum_only_ms, 4),
                             wide_second_pass_ms=round(msw / max(nw, 1), 4), eval_ms=round(mse / max(ne, 1), 4),
                             kernels="sep_accum_kernel + sep_scg_kernel + node_wide_sums / scan / eval (all passes over the table)"))
    # the whole recursion on this table (reference src/abawaca.cpp:95-197 with the GPU separator)
    t0 = time.perf_counter()
    nodes_all, s2c_all, _d2c = separator.binning(store)
    rec_s = time.perf_counter() - t0
    bins = [r for r in nodes_all if not r["split"]]
    # planted genomes vs bins: share of scaffolds whose bin's majority genome is their own
    lab = np.asarray(genome)
    order = np.argsort(s2c_all, kind="stable")
    cuts = np.flatnonzero(np.diff(s2c_all[order])) + 1
    agree = sum(int(np.bincount(g).max()) for g in np.split(lab[order], cuts) if len(g))
    res["recursion"] = dict(s=round(rec_s, 3), c_abi_s=round(sum(r["call_s"] for r in nodes_all), 3), nodes=len(nodes_all), bins=len(bins),
                            ms_per_node=round(rec_s / len(nodes_all) * 1e3, 3), planted_genomes=int(G),
                            scaffolds_in_their_bins_majority_genome=round(agree / len(lab), 4),
                            note="whole recursive binning of the table, GPU separator + Python work-list loop")
    # ... and the same recursion in ONE call: abw_sep_tree evaluates the nodes of a level together (static flip values,
    # smaller child accumulated / bigger child by subtraction, scan-based coverage dimensions)
    separator.binning_tree(store, max_nodes=3)                      # builds the static tables of the loaded table
    tree_s = None
    for _ in range(3):
        rows_t, s2c_t, dt = separator.binning_tree(store)
        tree_s = dt if tree_s is None else min(tree_s, dt)
    key = lambda r: (r["id"], r["ndps"], r["split"], r["dim"], r["key"], r["spec"], r["sens"], r["c1"], r["n1"], r["c2"], r["n2"])
    # per kernel family (HIP-event brackets inside the call) and against the bytes the recursion reads when every node reads
    # its own part of the table once, 4 bytes per (datapoint, dimension) -- SURVEY section 8(d)'s unit, summed over the nodes
    ctx.prof_reset(); ctx.prof_enable(True)
    separator.binning_tree(store)
    fam = {k: ctx.prof_read(k)[1] for k in ("sep_accum", "sep_wide", "sep_eval", "sep_partition")}
    ctx.prof_enable(False)
    node_bytes = 4.0 * D * sum(r["ndps"] for r in rows_t)
    res["recursion"]["one_call"] = dict(entry_point="abw_sep_tree", ms=round(tree_s * 1e3, 3), ms_per_node=round(tree_s / len(rows_t) * 1e3, 4),
                                        ms_by_family=dict(narrow_accumulate_and_eval=round(fam["sep_accum"], 3), wide_scans=round(fam["sep_wide"], 3),
                                                          winner_and_sides=round(fam["sep_eval"], 3), partition=round(fam["sep_partition"], 3)),
                                        roofline=dict(kernels="tree_* (all of the call)", bound="hbm", achieved=round(node_bytes / tree_s / 1e9, 1), peak=HBM_PEAK_GBS,
                                                      unit="GB/s", frac=round(node_bytes / tree_s / 1e9 / HBM_PEAK_GBS, 4), traffic=None,
                                                      algorithmic_bytes="4 B x dimensions x datapoints of every node, summed over the nodes"),
                                        identical_to_node_by_node=bool([key(r) for r in rows_t] == [key(r) for r in nodes_all]
                                                                       and np.array_equal(s2c_t, s2c_all)),
                                        speedup_over_node_by_node_c_abi=round(res["recursion"]["c_abi_s"] / tree_s, 2))
    del store
    return res


def configs2_on_one_gpu(ctx, dev, args):
    """configs[2]'s job -- 256 restarts on the 1M-scaffold table -- and the separator's recursion on that table, on ONE GPU:
    the base of the strong-scaling curve `bench.py --gpus N` measures."""
    import torch
    from abawaca_amd import kmeans
    cfg = CONFIGS[2]
    keys_t, dp2scaf, scaf_len, genome, G = make_points(20242, cfg["scaffolds"], cfg["samples"], dev, cfg["genomes"])
    D, N = keys_t.shape
    x_t = standardise(keys_t)
    R, iters = cfg["restarts"], args.iters
    chunk = kmeans.auto_chunk(ctx, N, R)
    t_in = torch.empty(R, dtype=torch.float64, device=dev)
    t_as = torch.empty((R, N), dtype=torch.uint8, device=dev)

    def step():
        kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R, 20240, iters, chunk, t_as, None, t_in)

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
    steps = max(1, min(args.steps, 2))
    elapsed, nl, ms = timed_steps(step, 1, steps, barrier, ctx)
    ms_avg = ms / max(nl, 1)
    tf = N * R * 4.0 * (D + 1) / (ms_avg * 1e-3) / 1e12
    res = dict(workload=f"configs[2]: {cfg['scaffolds']} scaffolds -> {N} dps x {D} dims, ONE job of {R} restarts x {iters} iterations, one GPU",
               value=N * R * iters * steps / elapsed, unit="point*iterations/s", ms_per_step=elapsed / steps * 1e3, chunk=chunk,
               roofline=dict(kernel="km_step_kernel", bound="mfma", achieved=round(tf, 3), peak=FP32_MFMA_PEAK_TFLOPS, unit="TFLOP/s",
                             frac=round(tf / FP32_MFMA_PEAK_TFLOPS, 4), ms_per_launch=round(ms_avg, 4), traffic=None))
    if not args.no_split:
        in_exact = t_in.clone()
        kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
        try:
            el_s, n_s, ms_s = timed_steps(step, 1, steps, barrier, ctx, prof_name="km_step_b3")
        finally:
            kmeans.set_arith(ctx, kmeans.ARITH_F32)
        res["km_split_bf16x3"] = dict(value=N * R * iters * steps / el_s, unit="point*iterations/s", ms_per_step=el_s / steps * 1e3,
                                      speedup_over_exact=round(elapsed / el_s, 3), km_step_b3_ms_per_launch=round(ms_s / max(n_s, 1), 4),
                                      same_best_restart=bool(int(t_in.argmin()) == int(in_exact.argmin())),
                                      worst_inertia_rel_diff=float(((t_in - in_exact).abs() / in_exact).max()))
    del t_as, x_t
    if not args.no_extras:
        keys_host = keys_t.cpu().numpy()
        del keys_t
        res["separator"] = separator_leg(ctx, keys_host, dp2scaf, scaf_len, genome, G, 20242, "configs[2]")
    return res


def configs4_on_one_gpu(ctx, dev, args):
    """configs[4]'s table (2M scaffolds, 20 samples, 400 planted genomes) on ONE GPU: the root node and the whole recursive
    binning, node by node and in one abw_sep_tree call -- the base of the scaffold-sharded runs of N > 1."""
    import torch
    keys_t, dp2scaf, scaf_len, genome, G = make_points(20245, args.sep_scaffolds, 20, dev, 400)
    keys_host = keys_t.cpu().numpy()
    del keys_t
    torch.cuda.empty_cache()
    res = separator_leg(ctx, keys_host, dp2scaf, scaf_len, genome, G, 20246, "configs[4]")
    res.pop("cpu_baseline", None)
    return res


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: this process has made no GPU call yet (counting devices does
    not initialise HIP on this image), so it starts N fresh children, one per GPU, with the launcher's environment
    (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), waits for all of them and exits with the first failure.  Rank 0 prints the
    JSON line on the stdout it inherits.  Nothing is re-exec'ed."""
    import torch
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        raise SystemExit("bench.py: no GPU visible")
    if args.gpus > n_dev and args.backend == "nccl":
        raise SystemExit(f"bench.py: --gpus {args.gpus} but only {n_dev} device(s) visible (RCCL needs one GPU per rank; "
                         f"--backend gloo lets the ranks share a GPU for a rehearsal)")
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ABW_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-u", os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            r = p.poll()
            if r is None:
                continue
            alive.remove(p)
            if r != 0 and rc == 0:
                rc = r if r > 0 else 1
                for q in alive:              # a rank died: its peers would wait in a collective for ever
                    q.terminate()
        time.sleep(0.05)
    raise SystemExit(rc)


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
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

    def barrier():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---------------- workload: configs[1] on one GPU, configs[2] on several ---------------------
    cfg_no = 1 if world == 1 else 2
    cfg = CONFIGS[cfg_no]
    args.scaffolds = args.scaffolds or cfg["scaffolds"]
    args.samples = args.samples or cfg["samples"]
    args.restarts = args.restarts or cfg["restarts"]
    args.genomes = args.genomes or (cfg["genomes"] if args.scaffolds == cfg["scaffolds"] else 0)
    keys_t, dp2scaf, scaf_len, genome, G = make_points(20240 + cfg_no, args.scaffolds, args.samples, dev, args.genomes)
    D, N = keys_t.shape
    x_t = standardise(keys_t)
    R_job, iters = args.restarts, args.iters
    legs = {}
    if world == 1:
        # ---- one GPU: the whole job
        R = R_job
        if args.chunk == 0:
            args.chunk = kmeans.auto_chunk(ctx, N, R)   # what the library itself picks for chunk = 0
        first_restart, seed = 0, 20240
        t_inertia = torch.empty(R, dtype=torch.float64, device=dev)
        t_cent = torch.empty((R, 2, D), dtype=torch.float32, device=dev)
        t_assign = torch.empty((R, N), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def step():
            kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R, seed, iters, args.chunk, t_assign, t_cent, t_inertia)
        elapsed, n_launch, ms_launch = timed_steps(step, args.warmup, args.steps, barrier, ctx)
        winner = multi.best_restart(t_inertia.cpu(), first_restart)
        value = N * R * iters * args.steps / elapsed
        kernel_N, kernel_R = N, R
        parallelism, scaling = "single GPU", "weak"
        # ---- the same job with the Lloyd passes on three bf16 planes (abw_km_set_arith, abw_km_b3.hip).  Not the headline:
        #      that arithmetic is held to the documented tolerance (tests/test_gpu_kmeans_lloyd.py), not to bit parity.
        km_split = None
        if not args.no_split:
            in_exact, a_exact = t_inertia.clone(), t_assign[int(winner[0])].clone()
            kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
            try:
                el_s, n_s, ms_s = timed_steps(step, args.warmup, args.steps, barrier, ctx, prof_name="km_step_b3")
            finally:
                kmeans.set_arith(ctx, kmeans.ARITH_F32)
            w_s = multi.best_restart(t_inertia.cpu(), first_restart)
            tf_s = N * R * 4.0 * (D + 1) / (ms_s / max(n_s, 1) * 1e-3) / 1e12
            split_traffic = None
            if (args.scaffolds, args.samples, R) == (200000, 10, 64):
                try:
                    split_traffic = json.load(open(os.path.join(ROOT, "profiles", "pmc_r03.json")))["km_step_b3_kernel"]["hbm_bytes_per_launch"]
                except Exception:
                    split_traffic = None
            km_split = dict(arith="x and D as three bf16 planes each, six plane products on v_mfma_f32_32x32x16_bf16, fp32 accumulation; "
                                  "final assign pass, reductions and update exact",
                            value=N * R * iters * args.steps / el_s, unit="point*iterations/s", ms_per_step=el_s / args.steps * 1e3,
                            speedup_over_exact=round(elapsed / el_s, 3),
                            roofline=dict(kernel="km_step_b3_kernel", bound="mfma", achieved=round(tf_s, 3), peak=FP32_MFMA_PEAK_TFLOPS,
                                          unit="TFLOP/s", frac=round(tf_s / FP32_MFMA_PEAK_TFLOPS, 4), ms_per_launch=round(ms_s / max(n_s, 1), 4),
                                          launches=n_s, traffic=split_traffic,
                                          note="priced like the exact kernel: the 4(d+1) fp32 flops per unit the step needs, against the fp32 "
                                               "matrix peak (frac > 1 is the point of the split); as executed: 18(d+1) bf16 flops per unit = "
                                               f"{round(4.5 * tf_s / 2500.0, 3)} of the 2.5 PFLOP/s bf16 peak; the kernel sits at the power limit (DESIGN 3.5b)"),
                            best_restart=dict(id=int(w_s[0]), inertia=float(w_s[1])),
                            same_best_restart=bool(int(w_s[0]) == int(winner[0])),
                            worst_inertia_rel_diff=float(((t_inertia - in_exact).abs() / in_exact).max()),
                            best_restart_points_in_other_cluster=int((t_assign[int(winner[0])] != a_exact).sum()))
            del in_exact, a_exact
        del t_assign
    else:
        # ---- leg A, "restarts": rank k owns restart ids [first, first + R_a) of the job's R_job, points replicated
        first_restart, R_a = multi.restart_range(R_job, rank, world)
        seed_a = multi.rank_seed(20240, first_restart)
        chunk_a = args.chunk or kmeans.auto_chunk(ctx, N, R_a)
        t_inertia = torch.empty(R_a, dtype=torch.float64, device=dev)
        t_cent = torch.empty((R_a, 2, D), dtype=torch.float32, device=dev)
        t_assign = torch.empty((R_a, N), dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def step_a():
            kmeans.kmeans_restarts_dev(ctx, x_t, N, D, R_a, seed_a, iters, chunk_a, t_assign, t_cent, t_inertia)
        el_a, nl_a, ms_a = timed_steps(step_a, args.warmup, args.steps, barrier, ctx)
        el_a = multi.max_elapsed(el_a, dist, dev)
        winner_a = multi.best_restart(t_inertia, first_restart, dist, dev)     # the job's result: (restart id, inertia, owner)
        dist.barrier()
        del t_assign
        legs["restarts"] = dict(value=N * R_job * iters * args.steps / el_a, ms_per_step=el_a / args.steps * 1e3, restarts_per_gpu=R_a,
                                points_per_gpu=N, chunk=chunk_a, km_step_ms=round(ms_a / max(nl_a, 1), 4),
                                collective="none on the data path; one gather of the inertias")
        # ---- leg B, "points": rank k holds points [b[k], b[k+1]) and runs all R_job restarts; per pass one exchange of the
        # cluster-1 partial sums [Rp][W] fp64.  The library's stream is torch's current stream for this leg, so its kernels,
        # the RCCL collectives and the adds are ordered on the device without host synchronisation.
        b = multi.shard_bounds(N, world)
        n_loc = b[rank + 1] - b[rank]
        x_loc = x_t[:, b[rank]:b[rank + 1]].contiguous()
        torch.cuda.synchronize()
        chunk_b = args.chunk or kmeans.auto_chunk(ctx, n_loc, R_job)
        t_a = torch.empty((R_job, n_loc), dtype=torch.uint8, device=dev)
        t_c2 = torch.empty((R_job, 2, D), dtype=torch.float32, device=dev)
        t_i2 = torch.empty(R_job, dtype=torch.float64, device=dev)
        ctx.use_torch_stream(args.backend == "nccl")

        def step_b():
            kmeans.kmeans_sharded_dev(ctx, x_loc, b[rank], N, R_job, 20240, iters, chunk_b, dist, t_a, t_c2, t_i2)
        el_b, nl_b, ms_b = timed_steps(step_b, args.warmup, args.steps, barrier, ctx)
        ctx.use_torch_stream(False)
        el_b = multi.max_elapsed(el_b, dist, dev)
        chk = t_i2.clone()
        multi._all_reduce(chk, dist.ReduceOp.MAX, dist)                      # every rank must hold the same inertia bits
        winner_b = multi.best_restart(t_i2.cpu(), 0)
        Wp = (D + 1 + 31) // 32 * 32
        legs["points"] = dict(value=N * R_job * iters * args.steps / el_b, ms_per_step=el_b / args.steps * 1e3, restarts_per_gpu=R_job,
                              points_per_gpu=int(n_loc), chunk=chunk_b, km_step_ms=round(ms_b / max(nl_b, 1), 4),
                              identical_on_all_ranks=bool(torch.equal(chk, t_i2)),
                              exchange_bytes_per_pass=int(8 * ((R_job + 31) // 32 * 32) * Wp), passes_per_step=iters + 1,
                              collective=f"all-gather + rank-ordered fp64 adds per pass over {args.backend}")
        # the same partition with the Lloyd passes on three bf16 planes (abw_km_set_arith): all R_job restarts are on every
        # GPU, so the split kernel applies whatever the world size; the exchange and the final pass are the exact path's
        if not args.no_split:
            in_exact = t_i2.clone()
            kmeans.set_arith(ctx, kmeans.ARITH_BF16X3)
            ctx.use_torch_stream(args.backend == "nccl")
            try:
                el_s, nl_s, ms_s = timed_steps(step_b, args.warmup, args.steps, barrier, ctx, prof_name="km_step_b3")
            finally:
                ctx.use_torch_stream(False)
                kmeans.set_arith(ctx, kmeans.ARITH_F32)
            el_s = multi.max_elapsed(el_s, dist, dev)
            legs["points_split_bf16x3"] = dict(value=N * R_job * iters * args.steps / el_s, ms_per_step=el_s / args.steps * 1e3,
                                               speedup_over_exact_points=round(el_b / el_s, 3), km_step_ms=round(ms_s / max(nl_s, 1), 4),
                                               same_best_restart=bool(int(t_i2.argmin()) == int(in_exact.argmin())),
                                               worst_inertia_rel_diff=float(((t_i2 - in_exact).abs() / in_exact).max()),
                                               note="not a candidate for `value`: held to the Lloyd tolerance, not to bit parity")
        del t_a, x_loc
        dist.barrier()
        best = "points" if legs["points"]["value"] >= legs["restarts"]["value"] else "restarts"
        value = legs[best]["value"]
        elapsed = legs[best]["ms_per_step"] * 1e-3 * args.steps
        if best == "points":
            n_launch, ms_launch, kernel_N, kernel_R, winner, args.chunk = nl_b, ms_b, int(n_loc), R_job, winner_b, chunk_b
        else:
            n_launch, ms_launch, kernel_N, kernel_R, winner, args.chunk = nl_a, ms_a, N, R_a, winner_a, chunk_a
        R = R_job
        parallelism, scaling = f"{best} x{world} (strong scaling of one {R_job}-restart job; both partitions in `partitions`)", "strong"

    # ---------------- multi-rank only: the old weak-scaling run (64 restarts per GPU on the configs[1] table) --------
    weak = None
    if dist is not None and not args.no_sep_sharded:
        c1 = CONFIGS[1]
        k1, _d, _l, _g, _G1 = make_points(20241, c1["scaffolds"], c1["samples"], dev, c1["genomes"])
        x1 = standardise(k1)
        D1, N1 = k1.shape
        R1 = c1["restarts"]
        f1, _n1 = multi.restart_range(world * R1, rank, world)
        ch1 = kmeans.auto_chunk(ctx, N1, R1)
        ti = torch.empty(R1, dtype=torch.float64, device=dev)

        def step_w():
            kmeans.kmeans_restarts_dev(ctx, x1, N1, D1, R1, multi.rank_seed(20240, f1), iters, ch1, None, None, ti)
        el_w, _nl, _ms = timed_steps(step_w, 1, args.steps, barrier, ctx)
        el_w = multi.max_elapsed(el_w, dist, dev)
        weak = dict(value=world * N1 * R1 * iters * args.steps / el_w, unit="point*iterations/s", scaling="weak", ms_per_step=el_w / args.steps * 1e3,
                    workload=f"configs[1] table ({N1} dps x {D1}), {R1} restarts per GPU = {world * R1} in total")
        del k1, x1
        dist.barrier()

    # ---------------- multi-rank only: the real separator with the SCAFFOLDS sharded over the ranks -----------
    # (SURVEY §8(e), configs[4]: 2M scaffolds in total, every rank holds 1/world of them -- generated from its own seed --
    # per node one integer all-reduce of the histograms + min/max of the SCG flip values.)  The recursion is run for
    # --sep-nodes nodes (0 = to the end).
    sep_sharded = None
    if dist is not None and not args.no_sep_sharded and args.sep_node_by_node:
        c4 = CONFIGS[4]
        s_loc = args.sep_scaffolds // world
        keys_k, dp2scaf_k, scaf_len_k, genome_k, G4 = make_points(20244 + 1000 * rank, s_loc, c4["samples"], dev, c4["genomes"])
        scg_scaf_k, scg_id_k = make_scgs(20244 + 1000 * rank, genome_k, G4)
        so_k, si_k = separator.scg_csr(scg_scaf_k, scg_id_k, s_loc)
        keys_k_host = keys_k.cpu().numpy()
        n_k = keys_k.shape[1]
        del keys_k
        t0 = time.perf_counter()
        dv, do = multi.global_dictionary(keys_k_host, dist, dev)
        dict_s = time.perf_counter() - t0
        sctx = _lib.Context(dev_index)
        t0 = time.perf_counter()
        sstore = separator.ShardedColumnStore(sctx, keys_k_host, dp2scaf_k, scaf_len_k, dv, do, so_k, si_k, 30)
        load_s = time.perf_counter() - t0
        nd = separator.separate_sharded(sstore, None, dist, dev)          # warm-up (RCCL channels, buffers)
        barrier()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            nd = separator.separate_sharded(sstore, None, dist, dev)
        torch.cuda.synchronize()
        sh_s = multi.max_elapsed((time.perf_counter() - t0) / reps, dist, dev)
        n_all = multi.sum_counts([n_k], dist, dev)[0]
        hist_words = int(nd.hist.numel())
        root = (int(nd.out.split), int(nd.out.dimension), int(nd.out.key))
        barrier()
        t0 = time.perf_counter()
        nodes_sh, _s2c = separator.binning_sharded(sstore, lambda ids: separator.separate_sharded(sstore, ids, dist, dev),
                                                   max_nodes=args.sep_nodes or (1 << 30))
        torch.cuda.synchronize()
        rec_s = multi.max_elapsed(time.perf_counter() - t0, dist, dev)
        sep_sharded = dict(value=D * n_all / sh_s, unit="dimension*points/s", ms_per_node=sh_s * 1e3, node="root of the union of the ranks' tables",
                           workload=f"configs[4] shape: {s_loc * world} scaffolds in total, {s_loc} per rank, {c4['samples']} coverage samples, 30 SCGs",
                           dps_total=int(n_all), split=root[0], dimension=root[1], key=root[2],
                           exchange_bytes_per_node=hist_words * 8, dictionary_s=round(dict_s, 2), load_s=round(load_s, 2),
                           recursion=dict(nodes=len(nodes_sh), s=round(rec_s, 3), ms_per_node=round(rec_s / max(1, len(nodes_sh)) * 1e3, 3),
                                          splits=int(sum(r["split"] for r in nodes_sh)), whole=bool(args.sep_nodes == 0 or len(nodes_sh) < args.sep_nodes),
                                          exchange_bytes_total=hist_words * 8 * len(nodes_sh)),
                           collective=f"all-reduce SUM int64 x{hist_words} per node over {args.backend}")
        del sstore, nd
        sctx.close()
        dist.barrier()

    # ---------------- multi-rank only: the separator with the SCAFFOLDS sharded, level by level ---------------------------
    # (configs[4]: "scaffold-sharded at 2/4/8 GPUs".  ONE table of --sep-scaffolds scaffolds, the same on every rank (same seed); rank r
    # keeps the signature columns of a contiguous range of scaffolds and its share of the coverage columns over all scaffolds
    # (abw_sep_stree_* + the level calls).  Per level: one integer all-reduce of the accumulated nodes' histogram slices, MIN / MAX
    # of the SCG lists, a record per node, a byte per scaffold.  Strong scaling of one recursion.)
    sep_stree = None
    if dist is not None and not args.no_sep_sharded:
        c4 = CONFIGS[4]
        keys_s, dp2scaf_s, scaf_len_s, genome_s, G4s = make_points(20244, args.sep_scaffolds, c4["samples"], dev, c4["genomes"])
        scg_scaf_s, scg_id_s = make_scgs(20244, genome_s, G4s)
        so_s, si_s = separator.scg_csr(scg_scaf_s, scg_id_s, args.sep_scaffolds)
        D4s, N4s = keys_s.shape
        colmax_s = keys_s.amax(dim=1).cpu().numpy()
        w_cols, n_cols = np.flatnonzero(colmax_s >= 1024), np.flatnonzero(colmax_s < 1024)
        bnd = separator.scaffold_ranges(np.bincount(dp2scaf_s, minlength=args.sep_scaffolds), world)
        p0, p1 = int(np.searchsorted(dp2scaf_s, bnd[rank])), int(np.searchsorted(dp2scaf_s, bnd[rank + 1]))
        mine_w = w_cols[rank::world]
        narrow_local = keys_s[torch.from_numpy(n_cols).to(dev)][:, p0:p1].cpu().numpy()
        wide_rows = keys_s[torch.from_numpy(mine_w).to(dev)].cpu().numpy() if len(mine_w) else None
        del keys_s
        torch.cuda.empty_cache()
        cn, cw = _lib.Context(dev_index), _lib.Context(dev_index)
        t0 = time.perf_counter()
        shard = separator.ScafShard(cn, cw, narrow_local, wide_rows, dp2scaf_s, scaf_len_s, so_s, si_s, 30, bnd[rank], bnd[rank + 1], mine_w, n_cols,
                                    colmax_s[n_cols] + 1)
        load_st = time.perf_counter() - t0
        del narrow_local, wide_rows
        separator.binning_stree(shard, dist, max_nodes=3)               # warm-up: static tables, RCCL channels
        barrier()
        best_st = None
        for _ in range(2):
            rows_st, s2c_st, dt_st, ex_st, lv_st = separator.binning_stree(shard, dist)
            dt_st = multi.max_elapsed(dt_st, dist, dev)
            best_st = dt_st if best_st is None else min(best_st, dt_st)
        chk = torch.tensor([len(rows_st), int(np.asarray(s2c_st, dtype=np.int64).sum() % (1 << 31))], dtype=torch.int64, device=dev)
        chks = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(chks, chk)
        sep_stree = dict(value=sum(r["ndps"] for r in rows_st) * D4s / best_st, unit="dimension*points/s (every node's datapoints x all dimensions)",
                         s=round(best_st, 4), nodes=len(rows_st), bins=int(sum(1 for r in rows_st if not r["split"])), levels=int(lv_st),
                         workload=f"configs[4]: ONE recursion on {args.sep_scaffolds} scaffolds -> {N4s} dps x {D4s} dims, 30 SCGs x {G4s} genomes",
                         scaffolds_per_rank=int(bnd[rank + 1] - bnd[rank]), signature_columns=int(len(n_cols)), coverage_columns_per_rank=int(len(mine_w)),
                         exchange_bytes_total=int(ex_st), exchange_bytes_per_level=int(ex_st // max(1, lv_st)), load_s=round(load_st, 2), scaling="strong",
                         same_tree_on_all_ranks=bool(all(torch.equal(c, chks[0]) for c in chks)),
                         collective=f"per level over {args.backend}: all-reduce SUM int64 of the accumulated nodes' histogram slices, MIN / MAX of the SCG lists, "
                                    f"all-gather of {separator._WIN_DTYPE.itemsize} B per node, all-reduce MAX of 1 B per scaffold")
        del shard
        cn.close(); cw.close()
        dist.barrier()

    # ---------------- multi-rank only: the separator with the DIMENSIONS spread over the ranks -------------------------
    # (configs[4]: ONE table of --sep-scaffolds scaffolds, the same on every rank (same seed); a rank loads all scaffolds but
    # only its share of the columns and the whole recursion runs level by level (abw_sep_tree_eval / assign / advance): per
    # level one record per node and one byte per scaffold cross the links.  Strong scaling of one recursion.)
    sep_dims = None
    if dist is not None and not args.no_sep_sharded:
        c4 = CONFIGS[4]
        keys_d, dp2scaf_d, scaf_len_d, genome_d, G4d = make_points(20244, args.sep_scaffolds, c4["samples"], dev, c4["genomes"])
        scg_scaf_d, scg_id_d = make_scgs(20244, genome_d, G4d)
        so_d, si_d = separator.scg_csr(scg_scaf_d, scg_id_d, args.sep_scaffolds)
        D4, N4 = keys_d.shape
        colmax = keys_d.amax(dim=1).cpu().numpy()
        wide_cols, narrow_cols = np.flatnonzero(colmax >= 1024), np.flatnonzero(colmax < 1024)
        my_dims = np.sort(np.concatenate([wide_cols[rank::world], narrow_cols[rank::world]])).astype(np.int64)
        keys_mine = keys_d[torch.from_numpy(my_dims).to(dev)].cpu().numpy()
        del keys_d
        torch.cuda.empty_cache()
        dctx = _lib.Context(dev_index)
        t0 = time.perf_counter()
        dstore = separator.DimShardStore(dctx, keys_mine, np.arange(len(my_dims)), dp2scaf_d, scaf_len_d, so_d, si_d, 30)
        dstore.dims = my_dims                                            # global column numbers of the rows loaded
        load_d = time.perf_counter() - t0
        del keys_mine
        separator.binning_tree_dims(dstore, dist, max_nodes=3)           # warm-up: static tables, RCCL channels
        barrier()
        best_d = None
        for _ in range(2):
            rows_d, s2c_d, dt_d, ex_d = separator.binning_tree_dims(dstore, dist, max_nodes=args.sep_nodes_dims)
            dt_d = multi.max_elapsed(dt_d, dist, dev)
            best_d = dt_d if best_d is None else min(best_d, dt_d)
        chk = torch.tensor([len(rows_d), int(np.asarray(s2c_d, dtype=np.int64).sum() % (1 << 31))], dtype=torch.int64, device=dev)
        chks = [torch.zeros_like(chk) for _ in range(world)]
        dist.all_gather(chks, chk)
        sep_dims = dict(value=sum(r["ndps"] for r in rows_d) * D4 / best_d, unit="dimension*points/s (every node's datapoints x all dimensions)",
                        s=round(best_d, 4), nodes=len(rows_d), bins=int(sum(1 for r in rows_d if not r["split"])),
                        workload=f"configs[4]: ONE recursion on {args.sep_scaffolds} scaffolds -> {N4} dps x {D4} dims, 30 SCGs x {G4d} genomes",
                        columns_per_rank=int(len(my_dims)), wide_columns_per_rank=int(len(wide_cols[rank::world])),
                        exchange_bytes_total=int(ex_d), load_s=round(load_d, 2), scaling="strong",
                        same_tree_on_all_ranks=bool(all(torch.equal(c, chks[0]) for c in chks)),
                        collective=f"per level: all-gather of {separator._WIN_DTYPE.itemsize} B per node, all-reduce MAX of 1 B per scaffold, over {args.backend}")
        del dstore
        dctx.close()
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
    achieved = kernel_N * kernel_R * flops_per_unit / (ms_avg * 1e-3) / 1e12      # per launch: the points and restarts ONE GPU's launch covers
    # HBM bytes per launch from the committed PMC passes (profiles/pmc_r03.json; collected on this
    # workload with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, corrected as the microarchitecture guide says)
    pmc_all = {}
    try:
        pmc_all = json.load(open(os.path.join(ROOT, "profiles", "pmc_r03.json")))
    except Exception:
        pmc_all = {}
    default_workload = world == 1 and (args.scaffolds, args.samples, R, args.sig_mbp) == (200000, 10, 64, 1000)
    traffic = pmc_all.get("km_step_kernel", {}).get("hbm_bytes_per_launch") if default_workload else None
    roofline = dict(kernel="km_step_kernel", bound="mfma", achieved=round(achieved, 3), peak=FP32_MFMA_PEAK_TFLOPS,
                    unit="TFLOP/s", frac=round(achieved / FP32_MFMA_PEAK_TFLOPS, 4), traffic=traffic,
                    launches=n_launch, launches_timed="every %d-th launch of the timed region is bracketed by HIP events" % PROF_PERIOD,
                    ms_per_launch=round(ms_avg, 4), flops_per_unit=flops_per_unit,
                    survey_flops_per_unit=7 * D,
                    achieved_at_survey_count=round(kernel_N * kernel_R * 7.0 * D / (ms_avg * 1e-3) / 1e12, 3),
                    hbm_algorithmic_GBs=round((4.0 * kernel_N * D + 4.0 * kernel_R * (D + 1)) / (ms_avg * 1e-3) / 1e9, 1),
                    per_launch=dict(points=int(kernel_N), restarts=int(kernel_R)))

    out = dict(metric="k-means point·iterations/sec (whole node) + signature build Gbp/s",
               value=value, unit="point*iterations/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
               ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling=scaling, vs_baseline=None,
               dtype="f32", data="synthetic",
               config=dict(workload=f"configs[{cfg_no}]: {args.scaffolds} scaffolds -> {N} dps, 179 signature + {args.samples} coverage dims "
                                    f"(z-scored), {G} planted genomes, k=2, ONE job of {R} restarts x {iters} Lloyd iterations, chunk {args.chunk}",
                           parallelism=parallelism),
               roofline=roofline, best_restart=dict(id=int(winner[0]), inertia=float(winner[1]), rank=int(winner[2])))
    if world == 1 and km_split is not None:
        out["km_split_bf16x3"] = km_split
    if legs:
        out["partitions"] = legs
    if weak is not None:
        out["weak_restarts"] = weak
    if sep_stree is not None:
        out["separator_sharded"] = sep_stree
    if sep_sharded is not None:
        out["separator_sharded_node_by_node"] = sep_sharded
    if sep_dims is not None:
        out["separator_dimension_sharded"] = sep_dims
    if sig_sharded is not None:
        out["signature_sharded"] = sig_sharded

    # ---------------- extras: real separator + signature build on this GPU -----------------------
    keys_host = None
    if not args.no_extras:
        keys_host = keys_t.cpu().numpy()
        out["separator"] = separator_leg(ctx, keys_host, dp2scaf, scaf_len, genome, G, 20240 + cfg_no, f"configs[{cfg_no}]",
                                         pmc_all.get("separator_root_node", {}).get("hbm_bytes_per_launch") if default_workload else None)

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
        # the boundary with HOST buffers (abw_sig_build: the caller's pageable ASCII through a ring of pinned staging
        # buffers, H2D / kernels / D2H overlapped chunk by chunk, milli rows back): PCIe-inclusive, never the headline
        out["signature"]["host_buffers"] = host_build_leg(ctx, blob, off)
        del t_dims
        if args.sig10_mbp and world == 1:
            out["signature"]["configs3_10gbp"] = signature_10g_leg(ctx, dev, args.sig10_mbp)
        t_dims = None
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
        # the same reads as a coordinate-sorted SAM file has them (another order of the additions: its own result)
        cov_unsorted = t_cov.clone()
        order = torch.argsort(r_scaf * 20_000 + r_pos, stable=True)
        t_rscaf, t_rpos = t_rscaf[order].contiguous(), t_rpos[order].contiguous()
        del order
        cov_step()
        ctx.prof_reset(); ctx.prof_enable(True)
        for _ in range(reps):
            cov_step()
        ncs, mscs = ctx.prof_read("cov_accumulate")
        ctx.prof_enable(False)
        # the two ways (pairs + stable sort; grouped reads summed where they lie) add the same contributions in another order
        cov_paths_agree = bool(torch.allclose(cov_unsorted, t_cov, rtol=1e-11, atol=1e-11) and float(t_cov.sum()) > 0)
        del cov_unsorted
        out["coverage"] = dict(value=n_reads / cov_s / 1e6, unit="M reads/s", reads=n_reads, segments=int(n_seg), ms_per_sample=cov_s * 1e3,
                               both_paths_agree_to_1e_11=cov_paths_agree,
                               coordinate_sorted=dict(ms_per_sample=round(mscs / ncs, 4), value=round(n_reads / (mscs / ncs * 1e-3) / 1e6, 1), unit="M reads/s",
                                                      frac=round(alg_cov / (mscs / ncs * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)),
                               note="row N1 (next): per-segment overlap sums in SAM order, fp64; reads and segments resident; "
                                    "the headline figures are for reads in NO order (the worst case: every contribution goes through a stable radix sort, "
                                    "hand-written, wave64 ballots, that brings each segment's contributions together in read order); "
                                    "`coordinate_sorted` is the same read set as a sorted SAM file has it (nothing is moved)",
                               roofline=dict(kernel="cov_accumulate (span, pairs, radix passes, sum)", bound="hbm", achieved=round(alg_cov / (mscv / ncv * 1e-3) / 1e9, 1),
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
        del t_ascii
        out["sam"] = side_leg("sam", sam_leg, ctx, dev, world == 1 and not args.no_cpu)
        out["lrn"] = side_leg("lrn", lrn_bench, ctx, world == 1 and not args.no_cpu)

    # ---------------- one GPU: configs[2]'s job and table on this GPU (the base of the N > 1 strong-scaling runs) ------
    x_host = x_t[:, :min(N, 200_000)].cpu().numpy() if (world == 1 and not args.no_cpu) else None
    if world == 1 and not args.no_c2:
        del x_t, keys_t
        x_t = keys_t = None
        torch.cuda.empty_cache()
        out["configs2_one_gpu"] = side_leg("configs2_one_gpu", configs2_on_one_gpu, ctx, dev, args)

    if world == 1 and not args.no_c4 and not args.no_extras:
        out["configs4_one_gpu"] = side_leg("configs4_one_gpu", configs4_on_one_gpu, ctx, dev, args)

    # ---------------- optional: configs[0], the reference's own CPU-runnable case, end to end -----------------
    if args.c1 and world == 1:
        out["recursion_c1"] = side_leg("recursion_c1", recursion_c1, ctx)

    # ---------------- CPU baselines (rank 0, N = 1 only) -------------------------------------------
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_kmeans_baseline(x_host, R, iters, args.chunk)
        out["cpu_baseline"]["gpu_over_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
        if not args.no_extras:
            b = side_leg("signature cpu_baseline", cpu_signature_baseline, blob, off)
            if b:
                out["signature"]["cpu_baseline"] = b
            b = side_leg("abawaca-build drop-in, end to end", build_dropin_leg, blob, off)
            if b:
                out["signature"]["dropin_end_to_end"] = b
            b = side_leg("separator cpu_baseline", cpu_separator_baseline, keys_host, dp2scaf, scaf_len)
            if b:
                out["separator"]["cpu_baseline"] = b
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
