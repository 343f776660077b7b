#!/bin/bash
# round 5, visit q: the column-sharing variant of the direct gather (-DMVS_GL_DIRECT16=4) against the product library: tests (bit-identity), per-launch times, whole path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
MVS_HIP_LIB=$L/libmvs_hip_direct4.so timeout 300 python -m pytest tests -m gpu -x -q -k "gather_variants or gather_windows or stage_lowp" 2>&1 | tail -2
for v in "" 4 "" 4; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_direct$v.so
    MVS_HIP_LIB="$lib" timeout 200 python scripts/prof_gather_direct.py 2>&1 | tail -3
done
B="--steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-shipped-leg"
for v in "" 4; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_direct$v.so
    MVS_HIP_LIB="$lib" timeout 300 python bench.py $B > gpurun_out/ab_q$v.json 2>gpurun_out/ab_q$v.err || tail -5 gpurun_out/ab_q$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/ab_q$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), '| fp16 tiles', round(r['fp16_tiles_handoff_mode']['value'],1))"
done
