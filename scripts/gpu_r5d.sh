#!/bin/bash
# Round 5, visit 4: weight-prefetch distance of the forward tile convolutions (MVS_WPF = 1 product, 2 / 3 / 4 variants) - the coarse stages'
# launches have no co-resident blocks to hide the per-step L2 latency of the packed weights behind.  Per-stage wall + whole path.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
for lib in mvsformerplusplus_amd/csrc/libmvs_hip.so mvsformerplusplus_amd/csrc/libmvs_hip_wpf*.so mvsformerplusplus_amd/csrc/libmvs_hip.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  echo "== $tag =="
  MVS_HIP_LIB=$PWD/$lib timeout 200 python scripts/diag_stage_time.py stagemix 2>&1 | grep -v amdgpu.ids
  MVS_HIP_LIB=$PWD/$lib timeout 200 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg > $OUT/ab.json 2>/dev/null
  python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$tag', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms; single', round(r['latency']['single_stream_ms_per_ref_view'],3))"
done
