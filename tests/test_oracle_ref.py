"""oracle/Makefile builds the reference twice from its own sources: with the flags of the reference's Makefile (no -O:
`*-asis`) and with -O2 (everyday use, CPU baselines).  The two builds must print the same bytes -- otherwise goldens made
with one would not pin the other.  Runs only where oracle/_ref exists (this container; nothing here needs a GPU)."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

needs_ref = pytest.mark.skipif(not all(os.path.exists(os.path.join(REF, x)) for x in
                                       ("abawaca-build", "abawaca-build-asis", "ref_sep_driver", "ref_sep_driver-asis")),
                               reason="oracle/_ref not built (make -C oracle ref)")


@needs_ref
def test_build_O0_and_O2_write_identical_files():
    from abawaca_amd import synth
    asm = synth.make_assembly(77, 12, 2, 2, frac_n_run=0.3)
    outs = {}
    with tempfile.TemporaryDirectory(prefix="abwref_") as wd:
        fasta = os.path.join(wd, "asm.fa")
        synth.write_fasta(fasta, asm)
        for s in range(2):
            synth.write_sam(os.path.join(wd, f"s{s}.sam"), asm, s, 500 + s, read_len=150)
        for exe in ("abawaca-build", "abawaca-build-asis"):
            out = os.path.join(wd, exe)
            os.makedirs(out)
            subprocess.run([os.path.join(REF, exe), "-f", fasta, "-o", out, "-s", os.path.join(wd, "s*.sam"), "-c", os.path.join(wd, "s0.sam")],
                           check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            outs[exe] = {k: open(os.path.join(out, "abawaca." + k), "rb").read() for k in ("lrn", "names", "info")}
    assert outs["abawaca-build"] == outs["abawaca-build-asis"]
    assert len(outs["abawaca-build"]["lrn"]) > 10000


@needs_ref
def test_separator_O0_and_O2_print_identical_trees(golden_dir):
    from abawaca_amd import synth
    g = np.load(os.path.join(golden_dir, "sep_g3.npz"))
    outs = {}
    with tempfile.TemporaryDirectory(prefix="abwref_") as wd:
        files = synth.write_build_dir(wd, g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]))
        for exe in ("ref_sep_driver", "ref_sep_driver-asis"):
            prefix = os.path.join(wd, exe)
            subprocess.run([os.path.join(REF, exe), files["names"], files["fasta"], files["info"], files["lrn"], files["scg"],
                            files["scg_list"], "4", prefix], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            nodes = ["\t".join(ln.split("\t")[:11]) for ln in open(prefix + ".nodes.tsv")]          # without the timing column
            outs[exe] = (nodes, open(prefix + ".scaf2cluster.tsv").read())
    assert outs["ref_sep_driver"] == outs["ref_sep_driver-asis"]
    assert len(outs["ref_sep_driver"][0]) >= 3
