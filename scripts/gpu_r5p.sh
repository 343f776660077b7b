#!/bin/bash
# round 5, visit p: sanity of the final library (gather tests, feature heads, smoke) + a short headline run
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 400 python -m pytest tests -m gpu -x -q -k "gather or feature_heads or stage_lowp or cascade_golden or fused_small" 2>&1 | tail -2
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-shipped-leg > gpurun_out/ab.json 2>gpurun_out/ab.err || tail -5 gpurun_out/ab.err
python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('headline', round(r['value'],1), '| fp16 tiles', round(r['fp16_tiles_handoff_mode']['value'],1), '| f16mix', round(r['uniform_f16mix_mode']['value'],1))"
