"""rocprofv3 --kernel-trace --stats database (rocpd) -> the per-kernel summary as text: calls, total, average, share.
usage: prof_summary.py results.db > summary.txt"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, total_calls, total_duration, average, percentage from top_kernels order by total_duration desc").fetchall()
print(f"{'kernel':90s} {'calls':>8s} {'total_ms':>12s} {'avg_us':>12s} {'share_%':>8s}")
for name, calls, tot, avg, pct in rows:
    short = name
    for junk in ("(anonymous namespace)::", "void "):
        short = short.replace(junk, "")
    short = short.split("(")[0]
    print(f"{short[:90]:90s} {calls:8d} {tot / 1e3:12.3f} {avg:12.2f} {pct:8.2f}")
