#!/usr/bin/env python3
"""Like pmc_summary.py, but keeps launches of different grid sizes apart (one kernel called on 2 000 and on 200 000 rows).
usage: pmc_by_grid.py <dir> [kernel-substring]"""
import csv
import glob
import re
import sys
from collections import defaultdict

d = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r.get("Kernel_Name", "")
        if "at::native" in name or "rocclr" in name or sub not in name:
            continue
        m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
        key = (m.group(1) if m else name[:50]) + " grid " + r.get("Grid_Size", "?")
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print(k, {c: (round(sum(v) / len(v), 1), len(v)) for c, v in acc[k].items()})
