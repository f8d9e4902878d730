"""The compiled drop-in (oracle/_ref/hip_sep_driver: the reference's main loop and containers + integration/ClusterSeparatorHip)
at a bench table's size, beside the reference's own separator on the root node.  Writes a JSON summary.
usage: python3 experiments/dropin_time.py [scaffolds] [samples] [genomes] [out.json] [ref_nodes]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from abawaca_amd import _lib, lrn

n_scaf = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
samples = int(sys.argv[2]) if len(sys.argv) > 2 else 10
genomes = int(sys.argv[3]) if len(sys.argv) > 3 else 100
out_json = sys.argv[4] if len(sys.argv) > 4 else os.path.join(ROOT, "gpurun_out", "dropin.json")
ref_nodes = int(sys.argv[5]) if len(sys.argv) > 5 else 1
with_node_by_node = (sys.argv[6] != "tree-only") if len(sys.argv) > 6 else True
dev = torch.device("cuda", 0)
keys_t, dp2scaf, scaf_len, genome, G = bench.make_points(20241, n_scaf, samples, dev, genomes)
D, N = keys_t.shape
scg_scaf, scg_id = bench.make_scgs(20241, genome, G)
ctx = _lib.Context(0)
res = dict(table=f"{n_scaf} scaffolds -> {N} dps x {D} dims, 30 SCGs x {G} genomes")
with tempfile.TemporaryDirectory(prefix="abwdrop_", dir="/tmp") as wd:
    t0 = time.perf_counter()
    with open(os.path.join(wd, "a.fa"), "wb") as f:
        for s in range(n_scaf):
            f.write(b">s%07d\n" % s + b"A" * int(scaf_len[s]) + b"\n")
    with open(os.path.join(wd, "a.info"), "w") as f:
        f.write("".join(f"s{s:07d}\t{int(scaf_len[s])}\t1.000\t0.500\t0\n" for s in range(n_scaf)))
    with open(os.path.join(wd, "a.names"), "w") as f:
        f.write(f"% {N}\n")
        f.write("".join(f"{p + 1}\ts{int(s):07d}_{p}\ts{int(s):07d}:(1, 2000), 2000/2000 non-Ns bps\n" for p, s in enumerate(dp2scaf)))
    with open(os.path.join(wd, "a.lrn"), "wb") as f:
        f.write((f"% {N}\n% {D + 1}\n% 9" + "\t1" * D + "\n% Key" + "".join(f"\td{j}" for j in range(D)) + "\n").encode())
        rows_t = keys_t.t().contiguous()
        step = 100_000
        for a in range(0, N, step):
            f.write(lrn.lrn_emit(ctx, rows_t[a:a + step].cpu().numpy(), first_key=a + 1).tobytes())
        del rows_t
    with open(os.path.join(wd, "genes.scg"), "w") as f:
        f.write("".join(f"s{int(s):07d}_{k + 1}\tSCG{int(g):03d}\n" for k, (s, g) in enumerate(zip(scg_scaf, scg_id))))
    with open(os.path.join(wd, "scg.list"), "w") as f:
        f.write("".join(f"SCG{g:03d}\n" for g in range(30)))
    res["write_files_s"] = round(time.perf_counter() - t0, 1)
    del keys_t
    ctx.close()
    torch.cuda.empty_cache()
    print("files written", res["write_files_s"], flush=True)
    files = [os.path.join(wd, x) for x in ("a.names", "a.fa", "a.info", "a.lrn", "genes.scg", "scg.list")]
    cores = min(40, int(os.environ.get("OMP_NUM_THREADS", "16")))

    def run(exe, tag, env=None, max_nodes=None):
        t0 = time.perf_counter()
        p = subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", exe)] + files + [str(cores), os.path.join(wd, tag)] + ([str(max_nodes)] if max_nodes else []),
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, env=dict(os.environ, **(env or {})))
        while p.poll() is None:                               # a line a minute: a silent run is taken for a hung one
            try:
                p.wait(timeout=45)
            except subprocess.TimeoutExpired:
                print(f"  {tag}: {time.perf_counter() - t0:.0f} s ...", flush=True)
        if p.returncode != 0:
            raise SystemExit(f"{exe} failed: {p.returncode}")
        wall = time.perf_counter() - t0
        rows = np.loadtxt(os.path.join(wd, tag + ".nodes.tsv"), ndmin=2)
        print(tag, "done", round(wall, 1), flush=True)
        return wall, rows

    w, tree = run("hip_sep_driver", "tree")
    stats = open(os.path.join(wd, "tree.sep_stats.tsv")).read().split()
    res["dropin_one_call_recursion"] = dict(wall_s=round(w, 1), nodes=len(tree), separate_total_s=round(float(tree[:, 11].sum()) / 1e6, 3),
                                            first_separate_s=round(float(tree[0, 11]) / 1e6, 3), later_separate_total_s=round(float(tree[1:, 11].sum()) / 1e6, 3),
                                            answered_from_tree=int(stats[0]), answered_node_by_node=int(stats[1]))
    if with_node_by_node:
        w, node = run("hip_sep_driver", "node", env={"ABW_NODE_BY_NODE": "1"})
        res["dropin_node_by_node"] = dict(wall_s=round(w, 1), nodes=len(node), separate_total_s=round(float(node[:, 11].sum()) / 1e6, 3),
                                          first_separate_s=round(float(node[0, 11]) / 1e6, 3), later_separate_total_s=round(float(node[1:, 11].sum()) / 1e6, 3))
        res["identical_nodes_and_bins"] = bool(np.array_equal(tree[:, :11], node[:, :11]) and
                                               open(os.path.join(wd, "tree.scaf2cluster.tsv")).read() == open(os.path.join(wd, "node.scaf2cluster.tsv")).read())
    if ref_nodes > 0:
        w, ref = run("ref_sep_driver", "ref", max_nodes=ref_nodes)
        res["reference"] = dict(wall_s=round(w, 1), nodes=len(ref), cores=cores, separate_total_s=round(float(ref[:, 11].sum()) / 1e6, 3),
                                root_separate_s=round(float(ref[0, 11]) / 1e6, 3), same_rows_as_dropin=bool(np.array_equal(ref[:, :11], tree[:len(ref), :11])),
                                note=f"oracle/_ref/ref_sep_driver (-O2), the first {ref_nodes} node(s) only")
    res["note"] = ("first_separate_s of the drop-in = the adapter building the library's table from the reference's Dimension objects "
                   "+ abw_sep_load + the whole recursion (abw_sep_tree) + the root's clusters; wall_s includes the reference's own "
                   "loaders (ScafDpData, ClusterData: the .lrn through atof and std::map), identical in all runs")
os.makedirs(os.path.dirname(out_json), exist_ok=True)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps(res))
