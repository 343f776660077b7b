#!/bin/bash
# round 6, visit j: do the four streams stay in lockstep?  One-off phase offsets between the streams (a throw-away `bench.py --stagger-ms` switch, removed after this visit: no effect, profiles/r06_batch_streams_ab.txt)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
B="--steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg"
for sg in 0 0.4 0.2 0.8 0 0.4; do
  echo "== stagger $sg ms =="
  timeout 300 python bench.py $B --stagger-ms $sg 2>gpurun_out/r6j_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(r[k],3) for k in ('value','ms_per_ref_view')})" || tail -3 gpurun_out/r6j_err.txt
done
