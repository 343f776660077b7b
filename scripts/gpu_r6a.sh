#!/bin/bash
# round 6, visit a: baseline of the round-5 library on this box + batch x streams A/B (batch the coarse stages)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
B="--steps 8 --warmup 2 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg"
for cfg in "1 4" "1 6" "2 2" "2 3" "4 1" "4 2" "2 4" "1 4"; do
  set -- $cfg
  echo "== batch $1 streams $2 =="
  timeout 300 python bench.py $B --batch $1 --streams $2 2>gpurun_out/r6a_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(r[k],3) for k in ('value','ms_per_ref_view')}, 'single-stream latency', round(r['latency']['single_stream_ms_per_ref_view'],3))" || tail -3 gpurun_out/r6a_err.txt
done
