#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest subset =="
timeout 900 python -m pytest tests -m gpu -x -q -k "vis_cnn or bd_hyp or stage_golden or cascade_golden or saturation" 2>&1 | tail -3
for st in 3 2 4 6; do
  echo "== bench streams=$st =="
  timeout 600 python bench.py --steps 8 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg --streams $st > $OUT/bench_s$st.json 2> $OUT/bench_s$st.err
  if [ $st = 3 ]; then grep -v "amdgpu.ids" $OUT/bench_s$st.err | head -12; fi
  python - $OUT/bench_s$st.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r['latency']['single_stream_ms_per_ref_view'], 'fam', {k: round(v['ms_per_ref_view'], 3) for k, v in r.get('families', {}).items()})
PY
done
echo "== bench shipped =="
timeout 900 python bench.py --steps 6 --warmup 2 --views-per-step 32 --profile-table --cost-reg shipped --no-cpu-baseline > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err
grep -v "amdgpu.ids" $OUT/bench_shipped.err | grep -E "tr_|pos3d|sum of"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench_shipped.json').read().strip().splitlines()[-1])
print('shipped', {k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r.get('latency', {}).get('single_stream_ms_per_ref_view'))
PY
