#!/bin/bash
# round 5, visit n: what bounds the direct gather?  product vs LDS windows vs two ablations (no interpolation / no loads), HIP-event times per launch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
for v in "" 0 2 3; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_direct$v.so
    MVS_HIP_LIB="$lib" timeout 200 python scripts/prof_gather_direct.py 2>&1 | tail -3
done
