#!/bin/bash
# GPU visit for the shipped regulariser mix (stage-1 transformer): parity tests + bench with per-kernel table.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python -m pytest tests -m gpu -x -q -k "transformer or shipped" 2>&1 | tail -6 | tee $OUT/pytest_gpu_tr.log
timeout 900 python bench.py --steps 5 --warmup 2 --profile-table --cost-reg shipped ${BENCH_ARGS:-} > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err
grep -v "amdgpu.ids" $OUT/bench_shipped.err | tail -50
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench_shipped.json').read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_step') if k in r}); print('cpu_baseline', r.get('cpu_baseline')); print('parity', r.get('parity'))
except Exception as e:
    print('bench_shipped.json unreadable', e)
PY
