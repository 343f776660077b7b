#!/bin/bash
# Round-3 GPU visit A: parity of the wave-autonomous gather kernels + A/B timing against the round-2 kernels (both pipelining depths).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu (gather / stage / cascade subset) =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "gather or stage or cascade or warp or cfg2" 2>&1 | tail -6 | tee $OUT/pytest_gpu_a.log
echo "== gather A/B: default library (MVS_GW_XUNIT=0) =="
timeout 600 python scripts/prof_gather.py --impls wave,lds 2>&1 | grep -v amdgpu.ids | tee $OUT/prof_gather_x0.log
echo "== gather A/B: cross-unit prefetch (MVS_GW_XUNIT=1) =="
MVS_HIP_LIB=$PWD/mvsformerplusplus_amd/csrc/libmvs_hip_x1.so timeout 600 python scripts/prof_gather.py --impls wave 2>&1 | grep -v amdgpu.ids | tee $OUT/prof_gather_x1.log
echo "== bench (default library) =="
timeout 900 python bench.py --steps 10 --warmup 3 --profile-table > $OUT/bench_a.json 2> $OUT/bench_a.err
grep -v "amdgpu.ids" $OUT/bench_a.err | tail -40
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench_a.json').read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_step', 'ms_per_ref_view') if k in r}); print('latency', r.get('latency')); print('parity', r.get('parity'))
except Exception as e:
    print('bench json unreadable', e)
PY
