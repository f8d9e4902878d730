import os, sys, json, subprocess
for v in ["1_2048", "1_8192", "4_8192", "2_4096", "4_16384"]:
    env = dict(os.environ, ABW_LIB_PATH=f"experiments/lib_sep_{v}.so")
    r = subprocess.run([sys.executable, "bench.py", "--no-cpu", "--steps", "1", "--iters", "1"], env=env, capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().split("\n")[-1])["separator"]
        print(v, d["ms_per_node"], d["roofline"]["ms_per_launch"])
    except Exception as e:
        print(v, "failed", r.stderr[-300:])
