#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests -m gpu -x -q -k "auto_policy or hip_graph" 2>&1 | tail -2
timeout 600 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg --conv-precision auto > gpurun_out/ab.json 2>gpurun_out/ab.err || tail -5 gpurun_out/ab.err
python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('auto', round(r['value'],1), 'ref-views/s', r['config']['conv_precision'][:140], '|', r['config']['issue'])"
