#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg --no-shipped-leg > gpurun_out/ab.json 2> gpurun_out/ab.err
grep -v amdgpu.ids gpurun_out/ab.err | tail -32
python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print(round(r['value'],1), {k:(round(v['ms_per_ref_view'],3)) for k,v in r['families'].items()}, 'roofline avg_launch_ms', r['roofline']['avg_launch_ms'], 'frac', r['roofline']['frac'])"
