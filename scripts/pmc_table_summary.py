#!/usr/bin/env python3
"""Same table as pmc_summary.py, from the merged table.txt written by pmc_table.py (when the raw CSVs were not kept):
   scripts/pmc_table_summary.py gpurun_out/pmc_<tag>/table.txt [name-regex] [launch-index]
With a launch index the per-launch value of that launch is used instead of the mean (first 16 launches are listed)."""
import re, sys
path = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""; idx = int(sys.argv[3]) if len(sys.argv) > 3 else None
ker = {}; cur = None
for line in open(path):
    if not line.startswith("    "):
        cur = line.strip(); ker[cur] = {}
        continue
    m = re.match(r"\s+(\S+)\s+n=(\d+)\s+mean\s+(\S+)\s+per-launch: (.*)", line)
    if m:
        per = [float(x) for x in m.group(4).split()]
        ker[cur][m.group(1)] = (float(m.group(3)) if idx is None or idx >= len(per) else per[idx], int(m.group(2)))
rows = []
for k, c in ker.items():
    if filt and not re.search(filt, k): continue
    m = {n: v[0] for n, v in c.items()}
    if m.get("SQ_WAVES", 0) == 0: continue
    w = m["SQ_WAVES"]; gui = m.get("GRBM_GUI_ACTIVE", 0) / 8.0; simd = gui * 1024.0; wc = m.get("SQ_WAVE_CYCLES", 0)
    pct = lambda x, y: 100.0 * x / y if y else float("nan")
    rows.append((gui, k, w, m.get("SQ_INSTS_VALU", 0) / w, m.get("SQ_INSTS_MFMA", 0) / w, m.get("SQ_INSTS_LDS", 0) / w, m.get("SQ_INSTS_SALU", 0) / w,
                 m.get("SQ_INSTS_VMEM_RD", 0) / w, pct(m.get("SQ_WAIT_ANY", 0), wc), pct(m.get("SQ_WAIT_INST_ANY", 0), wc), pct(m.get("SQ_ACTIVE_INST_ANY", 0), wc),
                 wc * 4 / simd if simd else float("nan"), pct(m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0), simd), pct(m.get("SQ_ACTIVE_INST_VALU", 0) * 4, simd),
                 pct(m.get("SQ_LDS_IDX_ACTIVE", 0), gui * 256.0), pct(m.get("SQ_LDS_BANK_CONFLICT", 0), m.get("SQ_LDS_IDX_ACTIVE", 0)), pct(m.get("TA_TA_BUSY", 0), gui * 256.0),
                 c.get("SQ_WAVES", (0, 0))[1]))
print("%-96s %7s %6s %5s %5s %5s %5s %5s | %5s %5s %5s | %4s | %5s %5s %5s %5s %5s | %3s" % ("kernel", "kcyc", "waves", "valu", "mfma", "lds", "salu", "vmem", "wait%", "stal%", "iss%", "w/S", "MFMA%", "VALU%", "LDS%", "conf%", "TA%", "n"))
for r in sorted(rows, reverse=True):
    print("%-96s %7.0f %6.0f %5.0f %5.0f %5.0f %5.0f %5.0f | %5.1f %5.1f %5.1f | %4.1f | %5.1f %5.1f %5.1f %5.1f %5.1f | %3d" % ((r[1][:96], r[0] / 1e3) + r[2:]))
