import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abawaca_amd import synth
KM = os.path.join(ROOT, "oracle", "_ref", "km_sep_driver")
for case in ["sep_g2noisy", "sep_g3", "sep_g3noscg", "sep_g4tight", "sep_g5", "sep_g6"]:
    g = np.load(os.path.join(ROOT, "tests", "golden", case + ".npz"))
    with tempfile.TemporaryDirectory(prefix="abwkm_") as wd:
        files = synth.write_build_dir(wd, g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]))
        prefix = os.path.join(wd, "km")
        r = subprocess.run([KM, files["names"], files["fasta"], files["info"], files["lrn"], files["scg"], files["scg_list"], "8", prefix],
                           capture_output=True, text=True, timeout=300)
        if r.returncode:
            print(case, "FAILED", r.stderr[-500:]); continue
        nodes = open(prefix + ".nodes.tsv").read().strip().split("\n")
        km = np.array([int(l.split("\t")[1]) for l in open(prefix + ".scaf2cluster.tsv").read().strip().split("\n")])
    ref = g["scaf2cluster"]
    ct = {}
    for a, b in zip(ref, km):
        ct[(int(a), int(b))] = ct.get((int(a), int(b)), 0) + 1
    print(case, "ref bins", len(set(ref)), "km bins", len(set(km)), "nodes", len(nodes), "contingency", sorted(ct.items()))
    for l in nodes[:3]: print("   ", l[:150])
