#!/bin/bash
# round 6, visit b: gather instruction diet (MVS_GL_OPT) - parity on the GPU, then same-box A/B against the round-5 form (libmvs_hip_glopt0.so)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gather or stage_golden or lowp or cascade_golden or cfg2" 2>&1 | tail -5
B="--steps 8 --warmup 2 --no-cpu-baseline --no-train-leg --no-shipped-leg --profile-table"
for v in "" glopt0 "" glopt0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    echo "=== variant [$v]"
    MVS_HIP_LIB="$lib" timeout 400 python bench.py $B > gpurun_out/r6b_$v.json 2> gpurun_out/r6b_$v.err || tail -5 gpurun_out/r6b_$v.err
    grep -E "^(gl_|corr_agg|vis_cnn|sum of)" gpurun_out/r6b_$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/r6b_$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), '| f16mix', round(r['uniform_f16mix_mode']['value'],1), '| bf16x3', round(r['fp32_equivalent_mode']['value'],1), '| tiles', round(r['fp16_tiles_handoff_mode']['value'],1))"
done
