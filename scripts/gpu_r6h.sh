#!/bin/bash
# round 6, visit h: staging rounds in flight (MVS_GL_SB = 2 product, 1 = round-5 order, 3) - parity, per-pass times, whole path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
L=$PWD/mvsformerplusplus_amd/csrc
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gather or stage_golden or lowp or cascade_golden or cfg2 or other_groups" 2>&1 | tail -4
timeout 300 python scripts/gather_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1
for v in glsb1 glsb3; do MVS_HIP_LIB=$L/libmvs_hip_$v.so timeout 300 python scripts/gather_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1; done
timeout 300 python scripts/gather_ablate.py 2>&1 | grep -v amdgpu.ids | tail -1
B="--steps 8 --warmup 2 --no-cpu-baseline --no-train-leg --no-shipped-leg --no-profile"
for v in "" glsb1 glsb3 "" glsb1 glsb3; do
    lib=""; [ -n "$v" ] && lib=$L/libmvs_hip_$v.so
    MVS_HIP_LIB="$lib" timeout 400 python bench.py $B > gpurun_out/r6h_$v.json 2> gpurun_out/r6h_$v.err || tail -5 gpurun_out/r6h_$v.err
    python -c "
import json; r = json.loads(open('gpurun_out/r6h_$v.json').read().strip().splitlines()[-1]); print('variant [$v] headline', round(r['value'],1), '| lat', round(r['latency']['single_stream_ms_per_ref_view'],3))"
done
