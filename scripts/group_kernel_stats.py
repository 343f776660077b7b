#!/usr/bin/env python3
"""Group a kernel-stats CSV (scripts/rocpd_stats.py) by __global__ FUNCTION - all template instantiations together, the tile and the
persistent form of the forward / transposed convolution together - the ranking bench.py's `roofline` object uses since round 5
(profiling.kernel_group):   python scripts/group_kernel_stats.py profiles/r05_kernel_stats.csv > profiles/r05_kernel_stats_by_function.csv
`AverageNs` of the row "conv3d_mfma_bf16x3_kernel" is what `roofline.avg_launch_ms` must agree with (one-stream run)."""
import csv
import re
import sys

agg = {}
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"^void ", "", r["Name"]).replace("mvs::", "")
    fn = re.sub(r"[<(].*$", "", name)
    fn = {"conv3d_mfma_bf16x3_persist_kernel": "conv3d_mfma_bf16x3_kernel", "deconv3d_mfma_bf16x3_persist_kernel": "deconv3d_mfma_bf16x3_kernel"}.get(fn, fn)
    a = agg.setdefault(fn, [0, 0, set()])
    a[0] += int(r["Calls"])
    a[1] += int(r["TotalDurationNs"])
    a[2].add(name.split("(")[0])
tot = sum(a[1] for a in agg.values()) or 1
print("Function,Calls,TotalDurationNs,AverageNs,Percentage,Instantiations")
for fn, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('"%s",%d,%d,%.1f,%.2f,%d' % (fn, a[0], a[1], a[1] / a[0], 100.0 * a[1] / tot, len(a[2])))
