#!/bin/bash
# round 6, visit d: z-padding step skipping (MVS_ZSKIP) - parity on the GPU, per-layer A/B against -DMVS_ZSKIP=0, and the first whole-path figure of the
# round-6 default policy ("auto") with its side legs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "single_layers or regnet or precisions or cascade_golden or cfg2 or stage_golden or auto_policy or vis" 2>&1 | tail -5
for v in "" zskip0 "" zskip0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    echo "=== variant [$v]"
    MVS_HIP_LIB="$lib" timeout 300 python scripts/bench_unet_layers.py 2>&1 | tail -22
done
B="--steps 8 --warmup 2 --no-cpu-baseline --no-train-leg --no-shipped-leg --profile-table"
for v in "" zskip0 ""; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    echo "=== bench variant [$v]"
    MVS_HIP_LIB="$lib" timeout 400 python bench.py $B > gpurun_out/r6d_$v.json 2> gpurun_out/r6d_$v.err || tail -5 gpurun_out/r6d_$v.err
    grep -E "^(sum of)" gpurun_out/r6d_$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/r6d_$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), r['config'].get('precision_policy'), '| exact coarse', round(r.get('exact_coarse_mode',{}).get('value',0),1), '| bf16x3', round(r['fp32_equivalent_mode']['value'],1), '| tiles', round(r.get('fp16_tiles_handoff_mode',{}).get('value',0),1), '| conv family', round(r['families']['regulariser_convolutions']['ms_per_ref_view'],3), '| lat', round(r['latency']['single_stream_ms_per_ref_view'],3))"
done
