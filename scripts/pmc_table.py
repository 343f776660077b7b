#!/usr/bin/env python3
"""Merge rocprofv3 counter_collection CSVs (one per --pmc pass) into a per-kernel table: scripts/pmc_table.py gpurun_out/pmc_x [filter]"""
import csv, glob, os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _demangle import demangle_mvs          # rocprofv3 leaves symbols with _Float16 parameters mangled
root = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for f in sorted(glob.glob(root + "/p*/p*_counter_collection.csv")):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(f)):
        name = demangle_mvs(row["Kernel_Name"])
        if filt and not re.search(filt, name): continue
        # operator + its whole template argument list (tile configuration included), namespaces and the parameter list dropped
        m = re.search(r"(warp_corr_\w+|conv3d_mfma\w*|deconv3d_mfma\w*|\w+_kernel)(<.*>)?\(", name)
        short = (m.group(1) + (m.group(2) or "").replace("mvs::", "").replace(" ", "")) if m else name[:50]
        per[short][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in per.items():
        for c, vals in cs.items():
            # one row per launch group: launches are kept in order, so same-shape launches (reps) can be told apart by the caller
            agg.setdefault(k, collections.OrderedDict())[c] = vals
cols = []
for k in agg:
    for c in agg[k]:
        if c not in cols: cols.append(c)
for k in agg:
    print(k)
    for c in cols:
        if c in agg[k]:
            v = agg[k][c]
            print("    %-34s n=%-3d mean %14.0f   per-launch: %s" % (c, len(v), sum(v) / len(v), " ".join("%.3g" % x for x in v[:16])))
