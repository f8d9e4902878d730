"""bench.py's output line (the driver's contract): one JSON object on stdout with the metric of BASELINE.json, the
timing fields, the roofline of the dominant kernel measured live and the CPU baseline beside it.  Runs last (file name)
because it is the slowest GPU test (the CPU baselines take about half a minute)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_run_prints_the_contract_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert d["metric"] == base["metric"] and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["dtype"] == "f32" and isinstance(d["config"], dict) and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 1e10 and d["ms_per_step"] > 0 and d["unit"] == "point*iterations/s"
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and 0 < r["frac"] <= 1 and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and (r["traffic"] is None or r["traffic"] > 0)
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    for leg in ("separator", "signature", "coverage", "lrn"):
        assert leg in d and d[leg]["value"] > 0
    sp = d["km_split_bf16x3"]                                    # the opt-in arithmetic beside the headline, never in its place
    assert sp["value"] > d["value"] and sp["same_best_restart"] and sp["worst_inertia_rel_diff"] <= 1e-6
    assert d["roofline"]["kernel"] == "km_step_kernel"


def test_bare_gpus_2_starts_its_own_ranks():
    """`bench.py --gpus 2` with no launcher around it: two ranks (sharing this box's one GPU over gloo), n_gpus == --gpus,
    configs[2]'s job under both partitions (at a reduced table so the test stays short)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "1", "--warmup", "1",
                        "--scaffolds", "40000", "--restarts", "64", "--iters", "5", "--sep-scaffolds", "40000", "--sep-nodes", "7",
                        "--sig-mbp", "200", "--cov-mreads", "1", "--no-cpu"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.strip().split("\n") if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and "configs[2]" in d["config"]["workload"]
    assert set(d["partitions"]) == {"restarts", "points", "points_split_bf16x3"} and d["partitions"]["points"]["identical_on_all_ranks"]
    sp = d["partitions"]["points_split_bf16x3"]                  # the opt-in arithmetic: never the headline, same winner, same inertia to 1e-6
    assert d["value"] in (d["partitions"]["restarts"]["value"], d["partitions"]["points"]["value"])
    assert sp["same_best_restart"] and sp["worst_inertia_rel_diff"] <= 1e-6
    assert d["separator_dimension_sharded"]["same_tree_on_all_ranks"]
    sh = d["separator_sharded"]                                  # scaffold-sharded, level by level: the same recursion, a small exchange
    assert sh["same_tree_on_all_ranks"] and sh["nodes"] == d["separator_dimension_sharded"]["nodes"]
    assert sh["exchange_bytes_per_level"] <= 8 << 20


def test_more_ranks_than_devices_is_refused():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode != 0 and not [ln for ln in p.stdout.split("\n") if ln.startswith("{")]
    assert "device(s) visible" in p.stderr
