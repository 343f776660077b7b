#!/bin/bash
# Round-2 visit A: parity of the LDS-staged gather kernels + per-stage A/B against the round-1 direct kernels + bench table.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.log
echo "== per-stage gather A/B =="
timeout 300 python scripts/prof_gather.py 2>&1 | grep -v amdgpu.ids | tee $OUT/prof_gather.log
echo "== bench (LDS-staged gather) =="
timeout 600 python bench.py --steps 20 --warmup 3 --profile-table --no-cpu-baseline > $OUT/bench_lds.json 2> $OUT/bench_lds.err
grep -v "amdgpu.ids" $OUT/bench_lds.err | tail -40
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench_lds.json').read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_step', 'hbm_algorithmic_frac_of_8TBs') if k in r})
PY
