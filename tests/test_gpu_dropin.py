"""The drop-in itself: the reference's own objects (ScafDpData, ClusterData, SCGdb, Cluster) driven by
the same work-list loop, once with the reference's ClusterSeparatorBySensitivitySpecificity and once
with integration/ClusterSeparatorHip calling libabawaca_hip.so.  Both drivers are built by
oracle/Makefile into oracle/_ref/ (from the reference's sources where they lie; nothing copied) and
travel to the GPU box as binaries.  Outputs must be identical text, timing column aside."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

from abawaca_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "ref_sep_driver")
HIP = os.path.join(ROOT, "oracle", "_ref", "hip_sep_driver")
HIP_SHARDED = os.path.join(ROOT, "oracle", "_ref", "hip_sharded_sep_driver")


def _run(exe, files, prefix, env=None, max_nodes=None):
    subprocess.run([exe, files["names"], files["fasta"], files["info"], files["lrn"], files["scg"], files["scg_list"],
                    "8", prefix] + ([str(max_nodes)] if max_nodes else []), check=True, timeout=300, env=env)
    nodes = [ln.split("\t")[:11] for ln in open(prefix + ".nodes.tsv").read().strip().split("\n")]
    # ... and the reference's summary.txt (bins with their sizes, GC, coverage, unique SCGs and average copies: the SCG
    # completeness / contamination columns), printed by the reference's own ClusterQuality in both drivers
    return nodes, open(prefix + ".scaf2cluster.tsv").read(), open(prefix + ".dp2cluster.tsv").read(), open(prefix + ".summary.txt").read()


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIP)), reason="oracle/_ref drivers not built")
@pytest.mark.parametrize("case", ["sep_g4tight", "sep_g3noscg", "sep_g6"])
def test_reference_main_loop_with_hip_separator(golden_dir, case):
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    with tempfile.TemporaryDirectory(prefix="abwdrop_") as wd:
        files = synth.write_build_dir(wd, g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]))
        ref = _run(REF, files, os.path.join(wd, "ref"))
        hip = _run(HIP, files, os.path.join(wd, "hip"))                    # nodes after the first are ClusterDataViews (row N2)
        hip_full = _run(HIP, files, os.path.join(wd, "hipf"), dict(os.environ, ABW_FULL_NODE_TABLES="1"))   # the reference's subset tables
    assert hip_full == hip
    assert ref[0] == hip[0]          # per node: id ndps split dim value spec sens children sizes (%.17g text)
    assert ref[1] == hip[1] and ref[2] == hip[2]
    assert ref[3] == hip[3] and ref[3].count("\n") == 1 + sum(1 for r in ref[0] if r[2] == "0")      # summary.txt: a row per bin
    # and the reference run on these regenerated files reproduces the committed golden nodes
    gold = g["nodes"]
    assert len(ref[0]) == len(gold)
    for row, r in zip(ref[0], gold):
        assert [int(row[0]), int(row[1]), int(row[2]), int(row[3])] == [int(r[0]), int(r[1]), int(r[2]), int(r[3])]
        assert float(row[4]) == r[4] and float(row[5]) == r[5] and float(row[6]) == r[6]


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIP)), reason="oracle/_ref drivers not built")
@pytest.mark.parametrize("case", ["sep_g4tight", "sep_g6"])
def test_cluster_files_written_by_the_gpu_writer_are_the_reference_files(golden_dir, case):
    """Row N4, output side of the drop-in: with ABW_CLUSTER_DIR set both drivers write the files of every cluster the
    recursion creates (reference src/abawaca.cpp:161-182) -- the reference through its ClusterEsomWriter, the GPU build
    through integration/ClusterEsomWriterHip (rows printed by abw_lrn_emit_rows from the resident table).  Every file
    must be byte-identical: <id>.lrn, <id>.scaf-stats.txt, <id>.scaf-cluster.txt."""
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    with tempfile.TemporaryDirectory(prefix="abwdropw_") as wd:
        files = synth.write_build_dir(wd, g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]))
        dirs = {}
        for tag, exe in (("ref", REF), ("hip", HIP)):
            dirs[tag] = os.path.join(wd, "clusters_" + tag)
            os.makedirs(dirs[tag])
            _run(exe, files, os.path.join(wd, tag), dict(os.environ, ABW_CLUSTER_DIR=dirs[tag]))
        names = sorted(os.listdir(dirs["ref"]))
        assert names == sorted(os.listdir(dirs["hip"])) and len(names) >= 6 and len(names) % 3 == 0
        n_rows = 0
        for nm in names:
            a = open(os.path.join(dirs["ref"], nm), "rb").read()
            b = open(os.path.join(dirs["hip"], nm), "rb").read()
            assert a == b, nm
            if nm.endswith(".lrn"):
                n_rows += a.count(b"\n") - 4
        assert n_rows > len(g["dp2scaf"]) // 2
        w_ref = open(os.path.join(wd, "ref.writer.tsv")).read().split()
        w_hip = open(os.path.join(wd, "hip.writer.tsv")).read().split()
        assert w_ref[1] == w_hip[1] == str(len(names) // 3)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIP_SHARDED)), reason="oracle/_ref drivers not built")
@pytest.mark.parametrize("case", ["sep_g4tight", "sep_g3noscg", "sep_g6"])
def test_reference_main_loop_with_the_sharded_hip_separator(golden_dir, case):
    """The single-process multi-GPU path: integration/ClusterSeparatorHipSharded keeps the table in
    ABW_SHARDS contexts (here all on the one device), accumulates every node on all of them, combines the histograms with
    abw_sep_shard_combine and lets every shard assign its own scaffolds.  Same text as the reference's separator, for 1, 2
    and 4 shards; the cluster files too."""
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    with tempfile.TemporaryDirectory(prefix="abwshard_") as wd:
        files = synth.write_build_dir(wd, g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]))
        ref_dir = os.path.join(wd, "clusters_ref")
        os.makedirs(ref_dir)
        ref = _run(REF, files, os.path.join(wd, "ref"), dict(os.environ, ABW_CLUSTER_DIR=ref_dir))
        for shards in (1, 2, 4):
            out_dir = os.path.join(wd, f"clusters_{shards}")
            os.makedirs(out_dir)
            got = _run(HIP_SHARDED, files, os.path.join(wd, f"sh{shards}"), dict(os.environ, ABW_SHARDS=str(shards), ABW_CLUSTER_DIR=out_dir))
            assert got[0] == ref[0], shards
            assert got[1] == ref[1] and got[2] == ref[2] and got[3] == ref[3], shards
            names = sorted(os.listdir(ref_dir))
            assert names == sorted(os.listdir(out_dir))
            for nm in names:
                assert open(os.path.join(ref_dir, nm), "rb").read() == open(os.path.join(out_dir, nm), "rb").read(), (shards, nm)


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(HIP)), reason="oracle/_ref drivers not built")
@pytest.mark.parametrize("case", ["sep_g2noisy", "sep_g5", "sep_g6"])
def test_the_adapter_answers_from_the_one_call_recursion(golden_dir, case):
    """integration/ClusterSeparatorHip runs abw_sep_tree at the root and serves the driver's later separate() calls from its
    records (+ abw_sep_node_assign for the raw sets and sides): every node must have been answered that way, with the text
    and the cluster files of the node-by-node path (ABW_NODE_BY_NODE=1) and of the reference; also when the driver cuts
    its work list."""
    g = np.load(os.path.join(golden_dir, case + ".npz"))
    with tempfile.TemporaryDirectory(prefix="abwdropt_") as wd:
        files = synth.write_build_dir(wd, g["milli"], g["dp2scaf"], g["scaf_len"], g["scg_scaf"], g["scg_id"], int(g["n_scg"]))
        out = {}
        for tag, exe, extra in (("ref", REF, {}), ("tree", HIP, {}), ("node", HIP, {"ABW_NODE_BY_NODE": "1"})):
            d = os.path.join(wd, "clusters_" + tag)
            os.makedirs(d)
            out[tag] = _run(exe, files, os.path.join(wd, tag), dict(os.environ, ABW_CLUSTER_DIR=d, **extra))
            out[tag + "_files"] = {nm: open(os.path.join(d, nm), "rb").read() for nm in sorted(os.listdir(d))}
        assert out["tree"] == out["node"] == out["ref"]
        assert out["tree_files"] == out["node_files"] == out["ref_files"]
        n_nodes = len(out["ref"][0])
        assert open(os.path.join(wd, "tree.sep_stats.tsv")).read().split() == [str(n_nodes), "0"]
        assert open(os.path.join(wd, "node.sep_stats.tsv")).read().split() == ["0", str(n_nodes)]
        if n_nodes > 3:
            cut_ref = _run(REF, files, os.path.join(wd, "cref"), max_nodes=3)
            cut_hip = _run(HIP, files, os.path.join(wd, "chip"), max_nodes=3)
            assert cut_ref == cut_hip and len(cut_hip[0]) == 3
