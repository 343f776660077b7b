#!/usr/bin/env python3
"""Compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one line per kernel
(VGPRs, AGPRs, SGPRs, scratch, occupancy).  Usage: scripts/kernel_resources.py <file.hip> [name filter] [extra hipcc flags...]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else ""
extra = [a for a in sys.argv[2:] if a.startswith("-")]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + extra
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT).stdout.decode()
if "error:" in out:
    print(out)
    sys.exit(1)
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line) or re.search(r" Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], stdout=subprocess.PIPE).stdout.decode().strip()}
        rows.append(cur)
        continue
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"TotalSGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"),
                     ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, line)
        if m and cur is not None:
            cur[key] = int(m.group(1))
for r in rows:
    if flt and flt not in r["name"]:
        continue
    name = re.sub(r"\(.*", "", r["name"]).replace("void mvs::", "")
    print("%-70s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d lds %d" % (name[:70], r.get("vgpr", -1), r.get("agpr", 0), r.get("sgpr", -1),
                                                                       r.get("scratch", 0), r.get("occ", -1), r.get("lds", 0)))
