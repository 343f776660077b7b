#!/bin/bash
# GPU visit 10: visibility CNN with the bias in the accumulator, packed conversions and a branch-free steady-state row loop
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest subset =="
timeout 900 python -m pytest tests -m gpu -x -q -k "vis_cnn or stage_golden or cascade_golden or wide_range or fullsize_properties" 2>&1 | tail -3
echo "== bench =="
timeout 600 python bench.py --steps 8 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg > $OUT/bench_v10.json 2> $OUT/bench_v10.err
grep -v "amdgpu.ids" $OUT/bench_v10.err | grep -E "vis_cnn|sum of" | head
python - $OUT/bench_v10.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r['latency']['single_stream_ms_per_ref_view'], 'fam', {k: round(v['ms_per_ref_view'], 3) for k, v in r.get('families', {}).items()})
print(r['roofline'])
PY
