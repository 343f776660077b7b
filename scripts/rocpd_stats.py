#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2) rocpd SQLite result as a kernel-stats CSV (what `--stats` prints):
   python scripts/rocpd_stats.py gpurun_out/prof/r01_results.db [substring] > profiles/r01_kernel_stats.csv
With a substring (e.g. "mvs::") only kernels whose name contains it are listed (percentages then refer to that subset):
bench.py generates its synthetic inputs with torch ops on the device, which a whole-process trace also records."""
import os
import re
import sqlite3
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _demangle import demangle_mvs          # rocprofv3 leaves symbols with _Float16 parameters mangled

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = db.execute("select %s, start, end from kernels" % name_col).fetchall()
agg = {}
for name, s, e in rows:
    name = demangle_mvs(re.sub(r"\s+", " ", name))
    a = agg.setdefault(name, [0, 0, 10 ** 18, 0])
    d = e - s
    a[0] += 1
    a[1] += d
    a[2] = min(a[2], d)
    a[3] = max(a[3], d)
if len(sys.argv) > 2:
    agg = {k: v for k, v in agg.items() if sys.argv[2] in k}
tot = sum(a[1] for a in agg.values()) or 1
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, a[2], a[3]))
