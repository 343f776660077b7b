#!/bin/bash
# round 5, visit r: same-box A/B of the direct gather (product) against the window form (-DMVS_GL_DIRECT16=0): per-launch times and the tiles leg of the bench
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
for v in "" 0 "" 0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_direct$v.so
    MVS_HIP_LIB="$lib" timeout 200 python scripts/prof_gather_direct.py 2>&1 | tail -3
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-shipped-leg"
for v in "" 0 "" 0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_direct$v.so
    MVS_HIP_LIB="$lib" timeout 300 python bench.py $B > gpurun_out/ab_r$v.json 2>gpurun_out/ab_r$v.err || tail -5 gpurun_out/ab_r$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/ab_r$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), '| fp16 tiles', round(r['fp16_tiles_handoff_mode']['value'],1), '| f16mix', round(r['uniform_f16mix_mode']['value'],1))"
done
