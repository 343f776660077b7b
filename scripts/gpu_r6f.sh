#!/bin/bash
# round 6, visit f: next-octet window prefetch in the C >= 32 gather passes (MVS_GL_PF) - parity, then same-box A/B against -DMVS_GL_PF=0
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gather or stage_golden or lowp or cascade_golden or cfg2 or other_groups" 2>&1 | tail -4
B="--steps 8 --warmup 2 --no-cpu-baseline --no-train-leg --no-shipped-leg --profile-table"
for v in "" glpf0 "" glpf0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    echo "=== variant [$v]"
    MVS_HIP_LIB="$lib" timeout 400 python bench.py $B > gpurun_out/r6f_$v.json 2> gpurun_out/r6f_$v.err || tail -5 gpurun_out/r6f_$v.err
    grep -E "^(gl_|corr_agg|sum of)" gpurun_out/r6f_$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/r6f_$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), '| exact coarse', round(r['exact_coarse_mode']['value'],1), '| bf16x3', round(r['fp32_equivalent_mode']['value'],1), '| tiles', round(r['fp16_tiles_handoff_mode']['value'],1), '| lat', round(r['latency']['single_stream_ms_per_ref_view'],3))"
done
