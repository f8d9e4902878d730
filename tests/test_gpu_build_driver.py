"""The compiled drop-in for `abawaca-build`: oracle/_ref/hip_build_driver = integration/abawaca_build_hip.cpp (the
reference's own FASTA reader; SAM records through abw_sam_parse with the reference's SAM reader, record parser and read
filter as the fall-back; segmentation + signatures, coverage sums and the rows of abawaca.lrn through the C ABI on the GPU)
linked with the reference's unmodified objects.  Its abawaca.lrn,
abawaca.names and abawaca.info must be the bytes the reference's own abawaca-build writes:

  * against the committed golden (tests/golden/build_b.npz: the files the reference wrote here, with the FASTA and SAM
    inputs, incl. records with 15 / 16 SNPs in their MD:Z and CIGAR strings on both sides of the max_snps rule), and
  * against oracle/_ref/abawaca-build (the reference compiled from its own sources) run beside it on a fresh assembly.
"""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from abawaca_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIP = os.path.join(ROOT, "oracle", "_ref", "hip_build_driver")
REF = os.path.join(ROOT, "oracle", "_ref", "abawaca-build")


def _run(exe, fasta, out, sam_glob, this_sam, env=None, stderr_has=None):
    os.makedirs(out, exist_ok=True)
    p = subprocess.run([exe, "-f", fasta, "-o", out, "-s", sam_glob, "-c", this_sam], capture_output=True, text=True, timeout=300,
                       env=None if env is None else {**os.environ, **env})
    assert p.returncode == 0, p.stderr[-2000:]
    assert stderr_has is None or stderr_has in p.stderr, p.stderr[-2000:]
    return {k: open(os.path.join(out, "abawaca." + k), "rb").read() for k in ("lrn", "names", "info")}


def test_driver_reproduces_the_golden_files(golden_dir):
    if not os.path.exists(HIP):
        pytest.fail("oracle/_ref/hip_build_driver is missing: build it with `make -C oracle ref` where /root/reference exists")
    z = np.load(os.path.join(golden_dir, "build_b.npz"))
    names = [str(n) for n in z["scaf_names"]]
    off = z["scaf_off"]
    with tempfile.TemporaryDirectory(prefix="abwdrv_") as wd:
        fasta = os.path.join(wd, "asm.fa")
        with open(fasta, "wb") as f:
            for i, n in enumerate(names):
                s = z["ascii"][off[i]:off[i + 1]].tobytes()
                f.write(b">" + n.encode() + b" some description\n")
                for o in range(0, len(s), 70):                                  # any line width: the reader joins the lines
                    f.write(s[o:o + 70] + b"\n")
        n_samples = len(z["read_off"]) - 1
        for s in range(n_samples):
            open(os.path.join(wd, f"s{s}.sam"), "wb").write(z[f"sam{s}"].tobytes())
        got = _run(HIP, fasta, os.path.join(wd, "out"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s0.sam"))
        mine = [os.path.join(wd, f"s{s}.sam").encode() for s in range(n_samples)]
    assert got["names"] == bytes(z["names"]) and got["info"] == bytes(z["info"])
    # abawaca.lrn: the '% Key' line names the SAM files by their paths (ref :511-512), which were another directory's
    want = bytes(z["lrn"]).split(b"\n")
    have = got["lrn"].split(b"\n")
    theirs = [str(n).encode() for n in z["sample_names"]]
    assert want[3].split(b"\t")[-n_samples:] == theirs and have[3].split(b"\t")[-n_samples:] == mine
    assert have[3].split(b"\t")[:-n_samples] == want[3].split(b"\t")[:-n_samples]
    assert have[:3] == want[:3] and have[4:] == want[4:] and len(have) > 50


@pytest.mark.parametrize("seed,n_scaf", [(91, 40)])
def test_driver_beside_the_reference_program(seed, n_scaf):
    if not (os.path.exists(HIP) and os.path.exists(REF)):
        pytest.fail("oracle/_ref/{hip_build_driver,abawaca-build} missing: build them with `make -C oracle ref`")
    asm = synth.make_assembly(seed, n_scaf, 3, 3, frac_n_run=0.4)
    asm.cov[:] = np.exp(synth.rng_for(seed + 1).normal(np.log(6.0), 0.3, size=asm.cov.shape))     # every segment sees reads
    with tempfile.TemporaryDirectory(prefix="abwdrv_") as wd:
        fasta = os.path.join(wd, "asm.fa")
        synth.write_fasta(fasta, asm)
        for s in range(3):
            p = os.path.join(wd, f"s{s}.sam")
            synth.write_sam(p, asm, s, seed + 10 + s, read_len=150)
            seq, q = "A" * 150, "I" * 150
            with open(p, "a") as f:                                             # dropped / kept by the reference's filter
                f.write(f"u0\t4\t*\t0\t0\t*\t*\t0\t0\t{seq}\t{q}\n")
                f.write(f"u1\t256\t{asm.names[0]}\t10\t255\t150M\t*\t0\t0\t{seq}\t{q}\n")
                f.write(f"u2\t0\tnot_a_scaffold\t10\t255\t150M\t*\t0\t0\t{seq}\t{q}\n")
                f.write(f"u3\t0\t{asm.names[1]}\t1\t255\t150M\t*\t0\t0\t{seq}\t{q}\tMD:Z:" + "8A" * 15 + "15\n")
                f.write(f"u4\t0\t{asm.names[1]}\t5\t255\t150M\t*\t0\t0\t{seq}\t{q}\tMD:Z:" + "8A" * 16 + "6\n")
                f.write(f"u5\t16\t{asm.names[2]}\t{len(asm.seqs[2]) - 30}\t255\t150M\t*\t0\t0\t{seq}\t{q}\n")
        ref = _run(REF, fasta, os.path.join(wd, "ref"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"))
        got = _run(HIP, fasta, os.path.join(wd, "hip"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"))
        old = _run(HIP, fasta, os.path.join(wd, "hip2"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"), env={"ABW_SAM_READER": "reference"})
        # the SAM files in pieces of 1 KiB (shorter than some lines: a piece grows until it holds a line end) and of 7 KiB
        cut1 = _run(HIP, fasta, os.path.join(wd, "hip4"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"), env={"ABW_SAM_PIECE_KB": "1"})
        cut7 = _run(HIP, fasta, os.path.join(wd, "hip5"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"), env={"ABW_SAM_PIECE_KB": "7"})
        # the assembly in batches of 16 KiB of sequence (a handful of scaffolds each; the default is 256 MiB): every batch is
        # segmented and counted by the worker thread while the reference's reader parses the next
        small = _run(HIP, fasta, os.path.join(wd, "hip6"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"), env={"ABW_BUILD_BATCH_KB": "16"})
        # a record abw_sam_parse refuses and the reference reads (atoi stops at the blank): that file falls back, the result stands
        with open(os.path.join(wd, "s1.sam"), "a") as f:
            f.write(f"u6\t0 \t{asm.names[3]}\t7\t255\t150M\t*\t0\t0\t{seq}\t{q}\n")
        ref3 = _run(REF, fasta, os.path.join(wd, "ref3"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"))
        got3 = _run(HIP, fasta, os.path.join(wd, "hip3"), os.path.join(wd, "s*.sam"), os.path.join(wd, "s1.sam"), stderr_has="FLAG or POS is not a plain number")
    assert got["lrn"] == ref["lrn"] and got["names"] == ref["names"] and got["info"] == ref["info"]
    assert old == got and cut1 == got and cut7 == got and small == got
    assert got3 == ref3 and got3["lrn"] != got["lrn"]
    assert got["lrn"].count(b"\n") > n_scaf
