#!/usr/bin/env python3
"""rocprofv3 PMC passes (FETCH_SIZE in <dir>/f, WRITE_SIZE in <dir>/w, made by scripts/gpu_round.sh) -> per-kernel HBM
traffic per launch:   python scripts/pmc_traffic.py gpurun_out/pmc_r01c profiles/pmc_traffic.json profiles/r01_pmc_fetch_write_raw.json
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: FETCH_SIZE counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md,
HBM section), so the read side is doubled."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _demangle import demangle_mvs
import csv
import glob
import json
import re
import sys


def load(dirname, counter):
    agg = {}
    for f in glob.glob(dirname + "/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            name = demangle_mvs(re.sub(r"\s+", " ", r["Kernel_Name"]))        # rocprofv3 leaves symbols with _Float16 parameters mangled
            name = re.sub(r"^void ", "", name).replace("mvs::", "")
            name = re.sub(r"\(.*$", "", name)
            a = agg.setdefault(name, [0, 0.0])
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


d, out, raw = sys.argv[1:4]
f, w = load(d + "/f", "FETCH_SIZE"), load(d + "/w", "WRITE_SIZE")
rawd, res = {}, {"_about": "HBM traffic per launch from rocprofv3 PMC (separate FETCH_SIZE and WRITE_SIZE passes of bench.py --steps 2 "
                           "--warmup 1, averaged over the launches of all stages). hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024: FETCH_SIZE "
                           "counts 64 B per 128-B request on gfx950 (MI355X_MICROARCH.md, HBM section) so the read side is doubled. These are the "
                           "L2's memory-side (fabric) requests: re-use absorbed by an XCD's L2 is excluded, Infinity-Cache hits are not."}
for k in sorted(set(f) | set(w)):
    if "at::native" in k or "elementwise" in k or k.startswith(("Cijk_", "__amd_")) or not k.strip():
        continue                                   # torch / rocBLAS kernels of the synthetic input generation, runtime copies
    fc, fv = f.get(k, [0, 0.0])
    wc, wv = w.get(k, [0, 0.0])
    fa, wa = fv / max(fc, 1), wv / max(wc, 1)
    rawd[k] = {"calls": max(fc, wc), "FETCH_SIZE_KB_avg": fa, "WRITE_SIZE_KB_avg": wa}
    short = re.sub(r"<.*$", "", k) if k.startswith("warp_corr") or k.startswith("weighted") else k
    res[short] = {"hbm_bytes_per_launch": (2 * fa + wa) * 1024, "fetch_size_kb": fa, "write_size_kb": wa, "launches_sampled": max(fc, wc)}
# whole path: every library kernel of the run / the reference views of the run (one cascade_prologue_kernel launch per view since round 5;
# confidence_average_kernel - once per view through round 4 - is fused into the last stage's head now)
views = max(f.get(k, [0])[0] for k in ("cascade_prologue_kernel", "confidence_average_kernel"))
views = max(views, max(w.get(k, [0])[0] for k in ("cascade_prologue_kernel", "confidence_average_kernel")))
if views:
    tot = sum((2 * f.get(k, [0, 0.0])[1] / max(f.get(k, [1])[0], 1) * max(f.get(k, [0])[0], w.get(k, [0])[0]) +
               w.get(k, [0, 0.0])[1] / max(w.get(k, [1])[0], 1) * max(f.get(k, [0])[0], w.get(k, [0])[0])) * 1024 for k in rawd
              if not any(t in k for t in ("bn_", "wgrad", "pack_", "_bwd", "train")))
    res["_whole_path"] = {"ref_views_in_run": views, "hbm_bytes_per_ref_view": tot / views,
                          "note": "sum over every library kernel launch of the PMC run of (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / reference views in the run"}
json.dump(res, open(out, "w"), indent=1)
json.dump(rawd, open(raw, "w"), indent=1)
print("wrote", out, raw, len(res) - 1, "kernels")
