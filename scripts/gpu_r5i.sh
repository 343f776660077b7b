#!/bin/bash
# streams 3 vs 4 in graph-replay mode, alternating (run-to-run spread is ~1 %)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
for args in "--streams 3" "--streams 4" "--streams 3" "--streams 4" "--streams 3" "--streams 4"; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg $args > gpurun_out/ab.json 2>/dev/null
  python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('%-12s' % '$args', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms')"
done
