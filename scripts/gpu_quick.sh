#!/bin/bash
# Short GPU visit: parity subset + bench with per-kernel table (+ optional counter list).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu =="
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== bench =="
timeout 600 python bench.py --steps 10 --warmup 2 --profile-table ${BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err
grep -v "amdgpu.ids" $OUT/bench.err | tail -45
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_step', 'hbm_algorithmic_frac_of_8TBs') if k in r})
    print('roofline', r.get('roofline')); print('cpu_baseline', r.get('cpu_baseline')); print('parity', r.get('parity'))
except Exception as e:
    print('bench.json unreadable', e)
PY
if [ "${1:-}" = "counters" ]; then rocprofv3 -L > $OUT/counters.txt 2>&1; wc -l $OUT/counters.txt; fi
