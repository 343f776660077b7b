#!/bin/bash
# Round-4 visit: per-layer A/B of the fp16 weight forms, parity subset, bench in the three fp16 formats.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== layers =="
timeout 300 python scripts/bench_layer.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r04_layer_ab.txt
echo "== pytest subset =="
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or golden or saturation or single_layers or midsize or cfg1" 2>&1 | tail -5 | tee $OUT/pytest_gpu_subset.log
for prec in f16mix f16 f16x2; do
  echo "== bench $prec =="
  timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg --conv-precision $prec > $OUT/bench_$prec.json 2> $OUT/bench_$prec.err
  grep -v "amdgpu.ids" $OUT/bench_$prec.err | grep -E "conv3d|deconv|sum of" | head -24
  python - $OUT/bench_$prec.json <<'PY'
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r['latency']['single_stream_ms_per_ref_view'], 'fam', {k: round(v['ms_per_ref_view'], 3) for k, v in r.get('families', {}).items()})
PY
done
