#!/bin/bash
# GPU visit 9: fp16 source windows in the LDS-staged gather (MVS_GATHER_F16) against fp32 windows (MVS_GATHER_WINDOW=f32)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest subset =="
timeout 900 python -m pytest tests -m gpu -x -q -k "gather or stage_golden or cascade_golden or saturation or cascade_vs_oracle" 2>&1 | tail -3
for mode in w16 w32; do
  [ $mode = w32 ] && export MVS_GATHER_WINDOW=f32
  echo "== bench $mode =="
  timeout 600 python bench.py --steps 8 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  grep -v "amdgpu.ids" $OUT/bench_$mode.err | grep -E "gl_|corr_agg|sum of" | head -14
  python - $OUT/bench_$mode.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r['latency']['single_stream_ms_per_ref_view'], 'fam', {k: round(v['ms_per_ref_view'], 3) for k, v in r.get('families', {}).items()})
PY
done
