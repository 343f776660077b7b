#!/bin/bash
# Round 5, visit 1: parity suite on the new default policy ("stagemix": exact coarse stages), bench in the new default (graph replay, shipped leg),
# A/B against round 4's uniform f16mix and against eager issue, error on the literal cfg4 / cfg5 range.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== bench (driver's command) =="
timeout 900 python bench.py --steps 20 --warmup 5 --profile-table > $OUT/bench.json 2> $OUT/bench.err
grep -v "amdgpu.ids" $OUT/bench.err | tail -48
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_step', 'ms_per_ref_view', 'dtype') if k in r})
    print('issue', r['config'].get('issue'))
    for k in ('latency', 'whole_path', 'roofline', 'cpu_baseline', 'parity', 'fp32_equivalent_mode', 'shipped'):
        print(k, r.get(k))
    print('families', {k: (v['ms_per_ref_view'], v.get('launches_per_ref_view')) for k, v in r.get('families', {}).items()})
except Exception as e:
    print('bench.json unreadable', e)
PY
for args in "--conv-precision f16mix" "--issue eager" ""; do
  echo "== A/B: bench.py --steps 10 $args =="
  timeout 600 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg $args > $OUT/ab.json 2>/dev/null
  python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('$args', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms; single', round(r['latency']['single_stream_ms_per_ref_view'],3), r['config']['issue'])"
done
echo "== literal cfg4 / cfg5 range =="
timeout 600 python scripts/diag_wide_range.py cfg4 2>&1 | grep -v amdgpu.ids
timeout 600 python scripts/diag_wide_range.py cfg5 2>&1 | grep -v amdgpu.ids
