#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (+ per-kernel HIP-event table), rocprofv3 kernel trace.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh [quick]'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== rocm-smi =="; rocm-smi --showproductname 2>/dev/null | head -8; nproc
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $OUT/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5 | tee $OUT/smoke.log
echo "== bench =="
timeout 900 python bench.py --steps 10 --warmup 2 --profile-table > $OUT/bench.json 2> $OUT/bench.err
tail -40 $OUT/bench.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_step', 'hbm_algorithmic_frac_of_8TBs') if k in r})
    print('roofline', r.get('roofline')); print('cpu_baseline', r.get('cpu_baseline')); print('parity', r.get('parity'))
except Exception as e:
    print('bench.json unreadable', e)
PY
if [ "${1:-}" != "quick" ]; then
  echo "== rocprofv3 kernel trace =="
  ROOT=$(pwd)
  cd /tmp && export TMPDIR=/tmp
  timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o r01 -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile > $ROOT/$OUT/rocprof.log 2>&1
  cd $ROOT
  tail -3 $OUT/rocprof.log
  find $OUT/prof -name '*kernel_stats*' | head; f=$(find $OUT/prof -name '*kernel_stats*.csv' | head -1); [ -n "$f" ] && head -30 "$f"
fi
