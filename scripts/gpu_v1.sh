#!/bin/bash
# Round-4 visit 1: pipe microbenchmarks, attention A/B, transformer parity subset, bench (all-Normal + shipped mix).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pipes2 =="
timeout 300 ./scripts/ubench/pipes2 > $OUT/r04_pipes2.txt 2>&1; tail -5 $OUT/r04_pipes2.txt
echo "== attention A/B =="
timeout 600 python scripts/prof_attn.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r04_attn_ab.txt
echo "== pytest subset =="
timeout 900 python -m pytest tests -m gpu -x -q -k "attention or transformer or shipped" 2>&1 | tail -5 | tee $OUT/pytest_gpu_subset.log
echo "== bench (all-Normal) =="
timeout 900 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg > $OUT/bench.json 2> $OUT/bench.err
grep -v "amdgpu.ids" $OUT/bench.err | tail -40
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_step', 'ms_per_ref_view') if k in r})
for k in ('latency', 'whole_path', 'roofline'):
    print(k, r.get(k))
PY
echo "== bench (shipped mix) =="
timeout 900 python bench.py --steps 4 --warmup 1 --views-per-step 32 --profile-table --cost-reg shipped > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err
grep -v "amdgpu.ids" $OUT/bench_shipped.err | grep -E "tr_|pos3d|softmax_regress|sum of"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench_shipped.json').read().strip().splitlines()[-1])
print('shipped', {k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r.get('latency'), 'parity', r.get('parity'))
PY
