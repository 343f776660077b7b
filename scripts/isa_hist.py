#!/usr/bin/env python3
"""Opcode-class histogram of the kernels of one csrc/*.hip: scripts/isa_hist.py <file.hip> <mangled-name substring> [extra flags]"""
import collections, os, re, subprocess, sys
src, flt = sys.argv[1], sys.argv[2]
extra = sys.argv[3:]
base = os.path.basename(src)[:-4]
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on", "-Wno-unused-value", "-save-temps",
                "-c", os.path.abspath(src), "-o", "/tmp/_isa.o"] + extra, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
s = open("/tmp/%s-hip-amdgcn-amd-amdhsa-gfx950.s" % base).read()
for f in re.split(r"\n(?=_Z\w+:)", s):
    m = re.match(r"(_Z\S+):", f)
    if not m or flt not in m.group(1):
        continue
    body = f.split(".section")[0]
    ops = collections.Counter()
    for line in body.splitlines():
        line = line.strip()
        if not line or line.startswith((".", ";", "_", "//")) or line.endswith(":"):
            continue
        ops[line.split()[0]] += 1
    g = collections.Counter()
    for op, c in ops.items():
        k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else op
        g[k] += c
    print(m.group(1), sum(ops.values()), dict(g))
    print("   ", {k: v for k, v in sorted(ops.items(), key=lambda kv: -kv[1]) if k.startswith(("ds_", "global_", "s_barrier", "v_rcp", "v_div_fmas", "v_exp", "v_log", "v_mfma", "s_waitcnt", "v_pk_fma", "v_fma", "v_cndmask", "v_mov"))})
