#!/usr/bin/env python3
"""Compact per-kernel pipe-utilisation table from the rocprofv3 counter CSVs of scripts/gpu_pmc_bench.sh / gpu_pmc.sh:
    scripts/pmc_summary.py gpurun_out/pmc_<tag> [name-regex]
Per kernel symbol (averaged over its launches): waves, instructions per wave (VALU / MFMA / LDS / SALU / VMEM), share of
wave-cycles spent waiting (s_waitcnt / barrier), stalled at issue, issuing; resident waves per SIMD; MFMA-pipe, VALU, LDS,
TA busy as a share of the kernel's duration (SIMD-cycles = GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs; SQ_* counters are quad-cycles
except SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES)."""
import collections, csv, glob, re, sys
root = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(root + "/p*/p*_counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        if filt and not re.search(filt, name):
            continue
        name = re.sub(r"^void (mvs::)?", "", name)
        name = re.sub(r"\(.*", "", name)
        vals[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
rows = []
for k, c in vals.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "SQ_WAVES" not in m or m["SQ_WAVES"] == 0:
        continue
    w = m["SQ_WAVES"]
    gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0
    simd_cyc = gui * 1024.0
    wc = m.get("SQ_WAVE_CYCLES", 0)
    pct = lambda x, y: 100.0 * x / y if y else float("nan")
    rows.append((gui, k, w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_MFMA", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SALU", 0) / w,
                 m.get("SQ_INSTS_VMEM_RD", 0) / w, pct(m.get("SQ_WAIT_ANY", 0), wc), pct(m.get("SQ_WAIT_INST_ANY", 0), wc), pct(m.get("SQ_ACTIVE_INST_ANY", 0), wc),
                 wc * 4 / simd_cyc if simd_cyc else float("nan"), pct(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), simd_cyc), pct(m.get("SQ_ACTIVE_INST_VALU", 0) * 4, simd_cyc),
                 pct(m.get("SQ_LDS_IDX_ACTIVE", 0), gui * 256.0), pct(m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 0)), pct(m.get("TA_TA_BUSY", 0), gui * 256.0 * 1.0),
                 len(c.get("SQ_WAVES", []))))
print("%-64s %7s %6s %5s %5s %5s %5s %5s | %5s %5s %5s | %4s | %5s %5s %5s %5s %5s | %3s" % ("kernel", "kcyc", "waves", "valu", "mfma", "lds", "salu", "vmem", "wait%", "stal%", "iss%", "w/S", "MFMA%", "VALU%", "LDS%", "conf%", "TA%", "n"))
for r in sorted(rows, reverse=True):
    print("%-64s %7.0f %6.0f %5.0f %5.0f %5.0f %5.0f %5.0f | %5.1f %5.1f %5.1f | %4.1f | %5.1f %5.1f %5.1f %5.1f %5.1f | %3d" % ((r[1][:64], r[0] / 1e3) + r[2:]))
