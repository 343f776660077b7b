#!/bin/bash
# round 6, visit k: wall time of the DEFAULT bench.py command (what the driver runs) and of its torch-ROCm composite leg under MIOpen's find modes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
t0=$(date +%s); timeout 1200 python bench.py > gpurun_out/r6k_a.json 2> gpurun_out/r6k_a.err; t1=$(date +%s); echo "default bench.py: $((t1-t0)) s wall"
python -c "
import json; r = json.loads(open('gpurun_out/r6k_a.json').read().strip().splitlines()[-1]); print(round(r['value'],1), r['torch_rocm_composite'].get('value'), r['torch_rocm_composite'].get('sample'))"
t0=$(date +%s); MIOPEN_FIND_MODE=FAST timeout 1200 python bench.py --steps 5 --no-train-leg --no-shipped-leg --no-profile > gpurun_out/r6k_b.json 2> gpurun_out/r6k_b.err; t1=$(date +%s); echo "MIOPEN_FIND_MODE=FAST short bench: $((t1-t0)) s wall"
python -c "
import json; r = json.loads(open('gpurun_out/r6k_b.json').read().strip().splitlines()[-1]); print(round(r['value'],1), r['torch_rocm_composite'].get('value'), r['torch_rocm_composite'].get('sample'))"
