"""Seeded synthetic inputs for tests, golden-vector generation and bench.py.

Follows the generator specified in SURVEY.md §8(d): G genomes with GC_g = linspace(0.30, 0.70, G),
per-sample coverage c_{g,s} = exp(N(ln 20, 1)), scaffold lengths 2000*n_s + U[0,1999] with
n_s in {2..6}, iid bases, one run of 50 'N' in 2 % of the scaffolds, 30 SCG names with 30 carrying
scaffolds per genome.  Everything is driven by numpy.random.Generator(PCG64(seed)).

Nothing here is on the product path: it only manufactures inputs (and the file formats the
reference's two executables exchange, reference src/abawaca-build.cpp:561-630).
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def rng_for(seed: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(seed))


@dataclass
class Assembly:
    names: list            # scaffold names (str)
    seqs: list             # np.uint8 arrays (ASCII)
    genome: np.ndarray     # genome id per scaffold
    gc: np.ndarray         # GC_g per genome
    cov: np.ndarray        # [G, S] coverage per genome and sample
    scg: list = field(default_factory=list)   # (gene_name, scg_name)


def make_assembly(seed: int, n_scaf: int, n_genomes: int, n_samples: int = 3,
                  n_lo: int = 2, n_hi: int = 6, frac_n_run: float = 0.02,
                  gc_lo: float = 0.30, gc_hi: float = 0.70, jitter: float = 0.004,
                  n_scg: int = 30, name_prefix: str = "scaffold") -> Assembly:
    rng = rng_for(seed)
    gc = np.linspace(gc_lo, gc_hi, n_genomes)
    cov = np.exp(rng.normal(np.log(20.0), 1.0, size=(n_genomes, n_samples)))
    genome = rng.integers(0, n_genomes, size=n_scaf)
    # make sure every genome has scaffolds when possible
    if n_scaf >= n_genomes:
        genome[:n_genomes] = np.arange(n_genomes)
    names, seqs = [], []
    for i in range(n_scaf):
        n_s = int(rng.integers(n_lo, n_hi + 1))
        length = 2000 * n_s + int(rng.integers(0, 2000))
        p_gc = float(np.clip(gc[genome[i]] + rng.normal(0.0, jitter), 0.05, 0.95))
        probs = np.array([(1 - p_gc) / 2, p_gc / 2, p_gc / 2, (1 - p_gc) / 2])
        s = ACGT[rng.choice(4, size=length, p=probs)]
        if rng.random() < frac_n_run and length > 200:
            at = int(rng.integers(0, length - 50))
            s = s.copy()
            s[at:at + 50] = ord("N")
        names.append(f"{name_prefix}_{i}")
        seqs.append(np.ascontiguousarray(s))
    scg = []
    for g in range(n_genomes):
        members = np.flatnonzero(genome == g)
        if len(members) == 0:
            continue
        carriers = rng.choice(members, size=min(n_scg, len(members)), replace=False)
        for j, sc in enumerate(carriers):
            scg.append((f"{names[sc]}_{1 + int(rng.integers(1, 50))}", f"SCG{j:02d}"))
    return Assembly(names, seqs, genome, gc, cov, scg)


def write_fasta(path: str, asm: Assembly, width: int = 60) -> None:
    with open(path, "wb") as f:
        for name, s in zip(asm.names, asm.seqs):
            f.write(b">" + name.encode() + b"\n")
            b = s.tobytes()
            for i in range(0, len(b), width):
                f.write(b[i:i + width] + b"\n")


def write_sam(path: str, asm: Assembly, sample: int, seed: int, read_len: int = 100,
              max_reads_per_scaf: int = 4000) -> int:
    """Single-end, perfectly matching reads (no MD tag -> 0 SNPs) at depth cov[g, sample]."""
    rng = rng_for(seed)
    n = 0
    qual = "I" * read_len
    with open(path, "w") as f:
        f.write("@HD\tVN:1.0\tSO:unsorted\n")
        for i, (name, s) in enumerate(zip(asm.names, asm.seqs)):
            depth = asm.cov[asm.genome[i], sample]
            L = len(s)
            if L <= read_len:
                continue
            n_reads = int(min(max_reads_per_scaf, rng.poisson(depth * L / read_len)))
            starts = rng.integers(0, L - read_len + 1, size=n_reads)
            for st in starts:
                seq = s[st:st + read_len].tobytes().decode()
                f.write(f"r{n}\t0\t{name}\t{st + 1}\t255\t{read_len}M\t*\t0\t0\t{seq}\t{qual}\n")
                n += 1
        if n == 0:   # the reference crashes on a SAM without alignment lines
            s = asm.seqs[0][:read_len].tobytes().decode()
            f.write(f"r0\t0\t{asm.names[0]}\t1\t255\t{len(s)}M\t*\t0\t0\t{s}\t{'I' * len(s)}\n")
            n = 1
    return n


def write_scg(path: str, asm: Assembly) -> None:
    with open(path, "w") as f:
        for gene, scg in asm.scg:
            f.write(f"{gene}\t{scg}\n")


def write_scg_list(path: str, n_scg: int = 30) -> None:
    with open(path, "w") as f:
        for j in range(n_scg):
            f.write(f"SCG{j:02d}\n")


# ------------------------------------------------------------------------------------------
# The file formats abawaca-build writes and abawaca reads (wire formats).
# ------------------------------------------------------------------------------------------
def read_lrn(path: str):
    """-> (dp_names int64[n], header list[str], milli int32[n, D]).  Values are the 3-decimal
    text of reference src/abawaca-build.cpp:603, carried exactly as integer thousandths."""
    with open(path) as f:
        lines = f.read().split("\n")
    header = lines[3].rstrip().split("\t")[1:]
    rows = [ln for ln in lines[4:] if ln and not ln.startswith("%")]
    n, D = len(rows), len(header)
    dp = np.empty(n, dtype=np.int64)
    milli = np.empty((n, D), dtype=np.int64)
    for i, ln in enumerate(rows):
        parts = ln.split("\t")
        dp[i] = int(parts[0])
        for j, t in enumerate(parts[1:]):
            neg = t.startswith("-")
            whole, _, frac = t.lstrip("-").partition(".")
            v = int(whole) * 1000 + int((frac + "000")[:3])
            milli[i, j] = -v if neg else v
    return dp, header, milli


def read_names(path: str):
    """-> list of (dp_name:int, segment_id:str, scaffold:str, start:int, end:int)."""
    out = []
    with open(path) as f:
        next(f)
        for ln in f:
            dp, seg, desc = ln.rstrip("\n").split("\t")
            scaf, rest = desc.split(":(", 1)
            a, b = rest.split(")", 1)[0].split(",")
            out.append((int(dp), seg, scaf, int(a), int(b)))
    return out


def read_info(path: str):
    out = {}
    with open(path) as f:
        for ln in f:
            p = ln.split()
            out[p[0]] = int(p[1])
    return out


def scafdp_numbering(names_rows):
    """Replays the id assignment of reference src/ScafDpData.cpp:56-99: scaffolds in byte-wise name
    order, scaffolds with exactly one dp dropped, dps of a scaffold in dp-name order, ids from 1.
    -> (kept_dp_names int64[N] in internal-id order, dp2scaf int32[N] (1-based), scaf_names list)"""
    by_scaf = {}
    for dp, _seg, scaf, _a, _b in names_rows:
        by_scaf.setdefault(scaf, set()).add(dp)
    kept, dp2scaf, scaf_names = [], [], []
    for scaf in sorted(by_scaf, key=lambda s: s.encode()):
        dps = sorted(by_scaf[scaf])
        if len(dps) == 1:
            continue
        scaf_names.append(scaf)
        for d in dps:
            kept.append(d)
            dp2scaf.append(len(scaf_names))
    return np.asarray(kept, dtype=np.int64), np.asarray(dp2scaf, dtype=np.int32), scaf_names


# ------------------------------------------------------------------------------------------
# Direct feature synthesis for the large separator / k-means configurations (no FASTA).
# ------------------------------------------------------------------------------------------
def canonical_mer_table():
    """(mer2dim int32[340], names list[180]): the mapping of reference
    src/abawaca-build.cpp:75-100, mers numbered k=1..4 blocks, base-4 with A,C,G,T = 0..3."""
    names, mer2dim, seen = [], [], {}
    for k in range(1, 5):
        for m in range(4 ** k):
            digs = [(m >> (2 * (k - 1 - i))) & 3 for i in range(k)]
            s = "".join("ACGT"[d] for d in digs)
            rc = "".join("TGCA"["ACGT".index(c)] for c in reversed(s))
            if rc in seen:
                seen[s] = seen[rc]
            else:
                seen[s] = len(names)
                names.append(s)
            mer2dim.append(seen[s])
    return np.asarray(mer2dim, dtype=np.int32), names


def make_features(seed: int, n_scaf: int, n_genomes: int, n_samples: int,
                  n_lo: int = 2, n_hi: int = 6, n_scg: int = 30, seg_len: float = 2500.0):
    """Milli-valued feature table shaped like abawaca.lrn (179 signature + S coverage columns).

    Signature columns: expected canonical k-mer frequency under iid bases with the scaffold's GC,
    plus binomial-scale noise for a ~2.5 kbp segment, truncated to 3 decimals as
    reference src/abawaca-build.cpp:603 does.  Coverage: c_{g,s} * (1 + N(0, 0.1)).
    -> dict(milli int32[D, N] column-major, dp2scaf int32[N] (0-based), scaf_ndps, scaf_len,
            genome int32[S#], scg_scaf int32[M], scg_id int32[M], n_scg)
    """
    rng = rng_for(seed)
    mer2dim, names = canonical_mer_table()
    gc = np.linspace(0.30, 0.70, n_genomes)
    cov = np.exp(rng.normal(np.log(20.0), 1.0, size=(n_genomes, n_samples)))
    genome = rng.integers(0, n_genomes, size=n_scaf).astype(np.int32)
    if n_scaf >= n_genomes:
        genome[:n_genomes] = np.arange(n_genomes)
    ndps = rng.integers(n_lo, n_hi + 1, size=n_scaf).astype(np.int32)
    scaf_len = (2000 * ndps + rng.integers(0, 2000, size=n_scaf)).astype(np.int64)
    N = int(ndps.sum())
    dp2scaf = np.repeat(np.arange(n_scaf, dtype=np.int32), ndps)
    p_gc = np.clip(gc[genome] + rng.normal(0, 0.004, size=n_scaf), 0.05, 0.95)[dp2scaf]
    D = 179 + n_samples
    milli = np.empty((D, N), dtype=np.int32)
    # per-mer expected frequency as a function of GC: (p/2)^(#GC) * ((1-p)/2)^(#AT)
    off = 0
    col_terms = [[] for _ in range(180)]
    for k in range(1, 5):
        for m in range(4 ** k):
            n_gc = sum(1 for i in range(k) if ((m >> (2 * i)) & 3) in (1, 2))
            col_terms[mer2dim[off + m]].append((n_gc, k - n_gc))
        off += 4 ** k
    for c in range(1, 180):
        f = np.zeros(N)
        for n_gc, n_at in col_terms[c]:
            f += (p_gc / 2) ** n_gc * ((1 - p_gc) / 2) ** n_at
        noise = rng.normal(0, 1, size=N) * np.sqrt(np.maximum(f * (1 - f), 0) / seg_len)
        milli[c - 1] = np.clip(np.floor((f + noise) * 1000.0), 0, 1000).astype(np.int32)
    for s in range(n_samples):
        c = cov[genome[dp2scaf], s] * (1.0 + rng.normal(0, 0.1, size=N))
        milli[179 + s] = np.floor(np.maximum(c, 0) * 1000.0).astype(np.int32)
    scg_scaf, scg_id = [], []
    order = np.argsort(genome, kind="stable")
    bounds = np.searchsorted(genome[order], np.arange(n_genomes + 1))
    for g in range(n_genomes):
        members = order[bounds[g]:bounds[g + 1]]
        if len(members) == 0:
            continue
        carriers = rng.choice(members, size=min(n_scg, len(members)), replace=False)
        scg_scaf.extend(int(x) for x in carriers)
        scg_id.extend(range(len(carriers)))
    return dict(milli=milli, dp2scaf=dp2scaf, scaf_ndps=ndps, scaf_len=scaf_len, genome=genome,
                scg_scaf=np.asarray(scg_scaf, dtype=np.int32), scg_id=np.asarray(scg_id, dtype=np.int32),
                n_scg=n_scg, names=names[1:] + [f"sample{s}" for s in range(n_samples)])


def write_build_dir(wd: str, milli, dp2scaf, scaf_len, scg_scaf=(), scg_id=(), n_scg=0):
    """Writes the files abawaca reads (reference src/abawaca.cpp:298-315) for a feature table:
    a.names / a.lrn / a.info / a.fa / genes.scg / scg.list.  Scaffold i is named s%07d so that the
    byte-wise name order of ScafDpData (reference src/ScafDpData.cpp:91-99) equals the index order."""
    milli = np.asarray(milli)
    D, N = milli.shape
    S = len(scaf_len)
    with open(os.path.join(wd, "a.fa"), "w") as f:
        for s in range(S):
            f.write(f">s{s:07d}\n" + "A" * int(scaf_len[s]) + "\n")
    with open(os.path.join(wd, "a.info"), "w") as f:
        for s in range(S):
            f.write(f"s{s:07d}\t{int(scaf_len[s])}\t1.000\t0.500\t0\n")
    with open(os.path.join(wd, "a.names"), "w") as f:
        f.write(f"% {N}\n")
        for p in range(N):
            s = int(dp2scaf[p])
            f.write(f"{p + 1}\ts{s:07d}_{p}\ts{s:07d}:(1, 2000), 2000/2000 non-Ns bps\n")
    with open(os.path.join(wd, "a.lrn"), "w") as f:
        f.write(f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\td{j}" for j in range(D)) + "\n")
        rows = milli.T
        for p in range(N):
            f.write(str(p + 1) + "".join("\t%d.%03d" % divmod(int(v), 1000) for v in rows[p]) + "\n")
    with open(os.path.join(wd, "genes.scg"), "w") as f:
        for k, (s, g) in enumerate(zip(scg_scaf, scg_id)):
            f.write(f"s{int(s):07d}_{k + 1}\tSCG{int(g):03d}\n")
    with open(os.path.join(wd, "scg.list"), "w") as f:
        for g in range(max(1, int(n_scg))):
            f.write(f"SCG{g:03d}\n")
    return dict(names=os.path.join(wd, "a.names"), fasta=os.path.join(wd, "a.fa"), info=os.path.join(wd, "a.info"),
                lrn=os.path.join(wd, "a.lrn"), scg=os.path.join(wd, "genes.scg") if len(scg_scaf) else "-",
                scg_list=os.path.join(wd, "scg.list"))
