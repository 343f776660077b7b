#!/bin/bash
# round 6, visit c: wave-autonomous visibility CNN (vis_cnn_wave_kernel) - parity on the GPU, then same-box A/B against the block form (MVS_VIS_BLOCK=1)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "vis or stage_golden or cascade_golden or cfg2" 2>&1 | tail -5
B="--steps 8 --warmup 2 --no-cpu-baseline --no-train-leg --no-shipped-leg --profile-table"
for v in "" 1 "" 1; do
    echo "=== MVS_VIS_BLOCK=[$v]"
    if [ -n "$v" ]; then export MVS_VIS_BLOCK=1; else unset MVS_VIS_BLOCK; fi
    timeout 400 python bench.py $B > gpurun_out/r6c_$v.json 2> gpurun_out/r6c_$v.err || tail -5 gpurun_out/r6c_$v.err
    grep -E "^(vis_cnn|sum of)" gpurun_out/r6c_$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/r6c_$v.json').read().strip().splitlines()[-1]); print('block=[$v] headline', round(r['value'],1), '| f16mix', round(r['uniform_f16mix_mode']['value'],1), '| bf16x3', round(r['fp32_equivalent_mode']['value'],1), '| tiles', round(r['fp16_tiles_handoff_mode']['value'],1), '| vis family', r['families']['visibility_cnn']['ms_per_ref_view'])"
done
