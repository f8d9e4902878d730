"""Per-kernel sums of the PMC counters in a rocprofv3 (rocpd) database.  usage: pmc_summary.py results.db [name filter]"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
view = "counters_collection" if "counters_collection" in tabs else None
if not view:
    print("tables:", tabs); sys.exit(1)
cols = [d[0] for d in c.execute(f"select * from {view} limit 1").description]
kn = "kernel_name" if "kernel_name" in cols else "name"
rows = c.execute(f"select {kn}, counter_name, count(*), sum(value) from {view} group by {kn}, counter_name").fetchall()
out = {}
for k, cn, n, v in rows:
    if flt and flt not in k: continue
    short = k.split("(")[0].split("::")[-1] if not k.startswith("(") else k.split("::")[1].split("(")[0]
    out.setdefault(short, {})[cn] = (n, v)
for k, d in sorted(out.items()):
    n = max(x[0] for x in d.values())
    print(k, f"dispatches~{n}")
    for cn, (m, v) in sorted(d.items()): print(f"    {cn:28s} {v:.4g}")
