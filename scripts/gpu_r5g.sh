#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests -m gpu -x -q -k "hip_graph" 2>&1 | tail -2
timeout 600 python scripts/track_s_bench.py 2>&1 | grep -v amdgpu.ids
for args in "" "--batch 2 --streams 2" "--batch 2 --streams 3"; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg $args > gpurun_out/ab.json 2>/dev/null
  python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('%-28s' % '$args', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms; single', round(r['latency']['single_stream_ms_per_ref_view'],3), r['config']['issue'])"
done
