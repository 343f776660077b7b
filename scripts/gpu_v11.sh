#!/bin/bash
# GPU visit 11: views per forward call x streams
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
for cfg in "1 3" "2 3" "2 2" "4 2" "3 3" "1 5"; do
  set -- $cfg
  timeout 600 python bench.py --steps 6 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --batch $1 --streams $2 > $OUT/bench_b$1s$2.json 2> $OUT/bench_b$1s$2.err
  python - $OUT/bench_b$1s$2.json "$cfg" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("batch/streams", sys.argv[2], {k: round(r[k], 3) for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', round(r['latency']['single_stream_ms_per_ref_view'], 3))
except Exception as e:
    print("batch/streams", sys.argv[2], "failed", e)
PY
done
