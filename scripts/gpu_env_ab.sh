#!/bin/bash
# GPU visit: parity subset, then bench (per-kernel table) of the product library under several environment settings.
# usage: PYTEST_K="..." TABLE_GREP="..." bash scripts/gpu_env_ab.sh "NAME=VALUE" "NAME=VALUE" ...   ("-" = unchanged environment)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu (subset: ${PYTEST_K:-all}) =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -4 | tee $OUT/pytest_gpu_ab.log
i=0
for setting in "$@"; do
    i=$((i+1))
    echo "== bench: $setting =="
    ( [ "$setting" != "-" ] && export "$setting"
      timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg ${BENCH_ARGS:-} > $OUT/bench_env$i.json 2> $OUT/bench_env$i.err )
    grep -v "amdgpu.ids" $OUT/bench_env$i.err | grep -E "${TABLE_GREP:-conv|deconv|sum of|kernel  }" | head -${TABLE_ROWS:-30}
    python - "$OUT/bench_env$i.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r.get('latency', {}).get('single_stream_ms_per_ref_view'), 'parity', r.get('parity', {}).get('refined_depth_rel_l1_vs_oracle'))
except Exception as e:
    print('bench json unreadable', e)
PY
done
