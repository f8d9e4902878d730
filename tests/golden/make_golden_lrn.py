#!/usr/bin/env python3
"""Golden vectors of the .lrn loader (row N4): a small feature table with every oddity the reference's loader
tolerates, read by the REFERENCE's own ClusterData constructor (oracle/_ref/lrn_dump, built by oracle/Makefile from
the sources under /root/reference; nothing of the reference is copied).  Output tests/golden/lrn_a.npz: the text of
the file we wrote (input) and the values the reference holds after loading it (expected output).

usage: python tests/golden/make_golden_lrn.py            (from the repo root)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from abawaca_amd import synth  # noqa: E402

rng = np.random.default_rng(11)
S, D = 6, 7
ndps_of = [2, 3, 2, 4, 2, 3]
dp2scaf = np.repeat(np.arange(S), ndps_of)
N = len(dp2scaf)
milli = rng.integers(0, 1001, size=(D, N))
milli[5] = rng.integers(0, 3_000_000, size=N)          # a coverage-like column
milli[6] = -rng.integers(0, 5000, size=N)              # negative values parse too (atof)
with tempfile.TemporaryDirectory(prefix="abwlrn_") as wd:
    files = synth.write_build_dir(wd, milli, dp2scaf, np.full(S, 9000))

    def fmt(v, style):
        sign = "-" if v < 0 else ""
        q, r = divmod(abs(int(v)), 1000)
        if style == 0:
            return f"{sign}{q}.{r:03d}"                 # what abawaca-build writes
        if style == 1:
            s = f"{sign}{q}.{r:03d}".rstrip("0")         # fewer decimals, possibly a bare "5."
            return s
        if style == 2:
            return f"{sign}{q}.{r:03d}000"               # more decimals, all zero
        return f"{sign}.{r:03d}" if q == 0 else f"{sign}{q}.{r:03d}"    # ".125"
    lines = [f"% {N + 1}", f"% {D + 1}", "% 9" + "\t1" * D, "% Key" + "".join(f"\tdim{j}" for j in range(D))]
    order = rng.permutation(N)                            # rows in any order
    for k, p in enumerate(order):
        row = str(p + 1) + "".join("\t" + fmt(milli[j, p], (k + j) % 4) for j in range(D))
        if k % 5 == 1:
            row += "  \t "                                # trailing blanks are chomped (ref :119-120)
        if k % 5 == 2:
            row += "\r"                                   # CRLF
        lines.append(row)
        if k == 3:
            lines.append("% a comment line")              # skipped (ref :123)
        if k == 6:
            lines.append("")                              # empty lines too
            lines.append("   ")
    lines.append("777" + "".join("\t1.000" for _ in range(D)))     # a datapoint the .names file does not know (ref :146-150)
    text = ("\n".join(lines)).encode()                    # no newline after the last row
    path = os.path.join(wd, "tricky.lrn")
    open(path, "wb").write(text)
    out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "lrn_dump"), files["names"], files["fasta"], files["info"], path, "1000"],
                         check=True, capture_output=True, text=True).stdout.strip().split("\n")
head = [int(x) for x in out[0].split("\t")]
rows = [ln.split("\t") for ln in out[1:]]
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lrn_a.npz"),
                    text=np.frombuffer(text, dtype=np.uint8), ndps_loaded=head[0], ndims=head[1],
                    names=np.array([int(r[0]) for r in rows]), iids=np.array([int(r[1]) for r in rows]),
                    values=np.array([[float(x) for x in r[2:]] for r in rows]))
print("reference loaded", head, "rows dumped", len(rows))


# ---- writer side: the abawaca.lrn the REFERENCE's abawaca-build printed for a small assembly with two samples ----
def writer_case():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    asm = synth.make_assembly(41, 12, 3, 2, frac_n_run=0.3)
    with tempfile.TemporaryDirectory(prefix="abwlrnw_") as wd:
        fasta, out_dir = mg.run_build(asm, wd, 2, 500)
        raw = open(os.path.join(out_dir, "abawaca.lrn"), "rb").read()
    lines = raw.split(b"\n")
    header, rows = lines[:4], [ln for ln in lines[4:] if ln]
    milli = np.array([[int(round(float(v) * 1000)) for v in ln.split(b"\t")[1:]] for ln in rows], dtype=np.int64)
    assert np.abs(milli).max() < 2 ** 31, "a coverage column holds the reference's out-of-bounds garbage; pick another seed"
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lrn_w.npz"), text=np.frombuffer(raw, dtype=np.uint8),
                        milli=milli.astype(np.int32), n_header_bytes=len(b"\n".join(header)) + 1)
    print("writer case:", len(rows), "rows x", milli.shape[1], "columns; coverage max", milli[:, 179:].max())


writer_case()


# ---- cluster writer: one cluster's .lrn as the REFERENCE's ClusterEsomWriter prints it ("%lu" + "\t%lf" per value,
#      ref src/ClusterWriter.cpp:92-98), for a table whose values need every kind of rounding to six decimals ----
def cluster_case():
    rng = np.random.default_rng(23)
    S, D = 20, 11
    ndps_of = rng.integers(2, 6, size=S)
    dp2scaf = np.repeat(np.arange(S), ndps_of)
    N = len(dp2scaf)
    special = ["0.0078125", "0.0234375", "-0.0078125", "999999.9999995", "0.9999995", "0.99999949", "1e-7", "-1e-9", "4.9e-324",
               "123456789.123456", "8999999999999.5", "0.0000005", "0.00000051", "-0", "0", "2.5", "1234.5678915", "0.1", "0.3",
               "7.0000004999", "99.9999995000001", "5e-7", "1.5e-6", "2.5e-6", "0.000244140625", "33554432.0000305"]
    cols = []
    for j in range(D):
        kind = j % 4
        if kind == 0:
            v = [f"{x:.3f}" for x in rng.random(N)]                              # what abawaca-build writes
        elif kind == 1:
            v = [repr(float(x)) for x in rng.random(N) * 10.0 ** rng.integers(-8, 9, size=N)]   # 17 significant digits
        elif kind == 2:
            v = [repr(float(x)) for x in -rng.random(N) * 1000]
        else:
            v = [special[(j * N + i) % len(special)] for i in range(N)]
        cols.append(v)
    k = 0
    for i in range(N):                                                            # every special value appears somewhere
        cols[3][i] = special[k % len(special)]; k += 1
    for i in range(N):                                                            # the cluster takes every second row
        cols[7][i] = special[(i // 2) % len(special)]
    with tempfile.TemporaryDirectory(prefix="abwlrnc_") as wd:
        files = synth.write_build_dir(wd, np.zeros((D, N), dtype=np.int64), dp2scaf, np.full(S, 9000))
        lines = [f"% {N}", f"% {D + 1}", "% 9" + "\t1" * D, "% Key" + "".join(f"\tdim{j}" for j in range(D))]
        for i in range(N):
            lines.append(str(i + 1) + "".join("\t" + cols[j][i] for j in range(D)))
        path = os.path.join(wd, "wild.lrn")
        open(path, "w").write("\n".join(lines) + "\n")
        cdir = os.path.join(wd, "clusters")
        os.makedirs(cdir)
        out = subprocess.run([os.path.join(ROOT, "oracle", "_ref", "lrn_dump"), files["names"], files["fasta"], files["info"], path, str(N), cdir, "2"],
                             check=True, capture_output=True, text=True).stdout.strip().split("\n")
        text = open(os.path.join(cdir, "c1.lrn"), "rb").read()
    rows = [ln.split("\t") for ln in out[1:]]
    iids = np.array([int(r[1]) for r in rows])
    values = np.zeros((D, N))                                                     # [dimension][internal dp id - 1], as the reference holds them
    for r in rows:
        values[:, int(r[1]) - 1] = [float(x) for x in r[2:]]
    body = text.split(b"\n", 4)[4]
    dps = np.array([int(ln.split(b"\t")[0]) for ln in body.split(b"\n") if ln], dtype=np.int64)   # the row order the reference chose
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lrn_c.npz"), values=values, dps=dps,
                        text=np.frombuffer(text, dtype=np.uint8), n_header_bytes=len(text) - len(body))
    print("cluster case:", len(dps), "of", N, "rows x", D, "columns,", len(body), "bytes of rows; loaded ids", len(iids))


cluster_case()
