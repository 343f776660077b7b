#!/bin/bash
# round 6, visit g: ablation of the gather unit (MVS_GL_ABL variants) - where the time of the passes goes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
timeout 300 python scripts/gather_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1
for n in 1 2 3 4 5 6; do
    MVS_HIP_LIB=$L/libmvs_hip_glabl$n.so timeout 300 python scripts/gather_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1
done
timeout 300 python scripts/gather_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1
