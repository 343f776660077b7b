#!/bin/bash
# Print VGPR / AGPR / scratch / occupancy of every kernel in one csrc file: scripts/kernel_resources.sh warp_kernels [filter]
f=$1; filt=${2:-.}
mkdir -p /tmp/kres && cd /tmp/kres
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on -Wno-pass-failed -Rpass-analysis=kernel-resource-usage -c /root/repo/mvsformerplusplus_amd/csrc/$f.hip -o $f.o 2> $f.res
python3 - "$f" "$filt" <<'PY'
import re, sys
txt = open('/tmp/kres/%s.res' % sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0].split(' [')[0]
    if not re.search(sys.argv[2], name): continue
    g = lambda k: int(re.search(k + r': (\d+)', b).group(1))
    print('%-100s vgpr %3d agpr %3d sgpr %3d scratch %4d occ %d' % (name[:100], g('VGPRs'), g('AGPRs'), g('SGPRs'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
PY
