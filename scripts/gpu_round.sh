#!/bin/bash
# Full GPU-box visit: parity tests, smoke, bench (+ per-kernel HIP-event table), rocprofv3 kernel trace, PMC traffic pass.
# Usage: gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh <tag>'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=gpurun_out; TAG=${1:-r01}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu =="
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
echo "== bench =="
timeout 900 python bench.py --steps 20 --warmup 3 --profile-table > $OUT/bench.json 2> $OUT/bench.err
grep -v "amdgpu.ids" $OUT/bench.err | tail -45
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_step', 'hbm_algorithmic_frac_of_8TBs') if k in r})
print('roofline', r.get('roofline')); print('cpu_baseline', r.get('cpu_baseline')); print('parity', r.get('parity'))
PY
echo "== bench, shipped regulariser mix (stage-1 transformer + PE3D) =="
timeout 900 python bench.py --steps 10 --warmup 2 --profile-table --cost-reg shipped > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err
grep -v "amdgpu.ids" $OUT/bench_shipped.err | grep -E "tr_|pos3d|softmax_regress|sum of"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench_shipped.json').read().strip().splitlines()[-1])
print('shipped', {k: r[k] for k in ('value', 'ms_per_step') if k in r}, 'parity', r.get('parity'))
PY
echo "== N = 2 flow of bench.py on this one-GPU box (both ranks on the device, gloo): code path check, not a measurement =="
MVS_BENCH_ONE_DEVICE=1 MVS_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 1 --no-profile > $OUT/bench_n2.json 2> $OUT/bench_n2.err
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench_n2.json').read().strip().splitlines()[-1])
    print('n2', {k: r[k] for k in ('value', 'n_gpus', 'ms_per_step')}, r.get('view_sharded'))
except Exception as e:
    print('bench_n2.json unreadable', e)
PY
echo "== rocprofv3 kernel trace (same bench command, 5 steps) =="
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile > $ROOT/$OUT/rocprof.log 2>&1
tail -2 $ROOT/$OUT/rocprof.log
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_${TAG}_shipped -o ${TAG}_shipped -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-profile --cost-reg shipped > $ROOT/$OUT/rocprof_shipped.log 2>&1
tail -1 $ROOT/$OUT/rocprof_shipped.log
echo "== PMC: HBM traffic of every kernel (separate passes) =="
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_$TAG/f -o f -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $ROOT/$OUT/pmc_f.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_$TAG/w -o w -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $ROOT/$OUT/pmc_w.log 2>&1
cd $ROOT; find $OUT/pmc_$TAG -name '*counter_collection.csv' | head
