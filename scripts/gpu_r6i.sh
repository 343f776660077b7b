#!/bin/bash
# round 6, visit i: one-tile transposed convolutions (fp16 formats) with all tile loads issued back to back (MVS_UNROLL_STAGE_F16) vs the rolled staging loop
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "single_layers or regnet or precisions or cascade_golden or cfg2 or gather" 2>&1 | tail -4
for v in "" dus0 "" dus0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    echo "=== variant [$v]"
    MVS_HIP_LIB="$lib" timeout 300 python scripts/bench_unet_layers.py 2>&1 | grep -E "deconv|sum"
done
B="--steps 8 --warmup 2 --no-cpu-baseline --no-train-leg --no-shipped-leg --no-profile"
for v in "" dus0 "" dus0; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    MVS_HIP_LIB="$lib" timeout 400 python bench.py $B > gpurun_out/r6i_$v.json 2> gpurun_out/r6i_$v.err || tail -5 gpurun_out/r6i_$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/r6i_$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), '| lat', round(r['latency']['single_stream_ms_per_ref_view'],3))"
done
