"""Per-kernel summary of a rocprofv3 run that left a rocpd sqlite file (`rocprofv3 --kernel-trace --stats -d DIR -o NAME`).
usage: python3 tools/rocpd_stats.py DIR/NAME_results.db [substring ...]   -> name, calls, total ms, average us"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = c.execute(f"select {name}, count(*), sum(end - start) / 1e6, avg(end - start) / 1e3 from kernels group by {name} order by 3 desc").fetchall()
pats = sys.argv[2:]
print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s}")
for n, k, tot, avg in rows:
    short = n
    for pre in ("void ", "(anonymous namespace)::"):
        short = short.replace(pre, "")
    short = short.split("(")[0][:70]
    if pats and not any(p in n for p in pats):
        continue
    print(f"{short:70s} {k:7d} {tot:10.3f} {avg:10.2f}")
