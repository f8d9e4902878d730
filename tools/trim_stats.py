#!/usr/bin/env python3
"""Trims a rocprofv3 *_kernel_stats.csv to this library's kernels with short names.
usage: trim_stats.py in.csv out.csv"""
import csv
import re
import sys

rows = list(csv.reader(open(sys.argv[1])))
out = [rows[0]]
for r in rows[1:]:
    name = r[0]
    if "at::native" in name or "rocclr" in name:
        continue
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    r[0] = m.group(1) if m else name[:60]
    out.append(r)
csv.writer(open(sys.argv[2], "w")).writerows(out)
for r in out:
    print(",".join(r))
