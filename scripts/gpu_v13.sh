#!/bin/bash
# Stream-count re-check on the round-4 kernels (round 2 chose 3 streams at 2.26 ms per view; the path is at 1.5 ms now).
# Usage: gpurun --timeout 240 -- 'bash scripts/gpu_v13.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
: > $OUT/streams_ab_v13.txt
for cfg in "--streams 3" "--streams 2" "--streams 4" "--streams 6" "--streams 3 --batch 2" "--streams 3"; do
  timeout 100 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg $cfg > $OUT/b13.json 2> $OUT/b13.err
  python - "$cfg" <<'PY' | tee -a $OUT/streams_ab_v13.txt
import json, sys
try:
    r = json.loads(open('gpurun_out/b13.json').read().strip().splitlines()[-1])
    print("%-24s %7.1f ref-views/s  %6.3f ms/view  single-stream %6.3f ms" % (sys.argv[1], r['value'], r['ms_per_ref_view'], r['latency']['single_stream_ms_per_ref_view']))
except Exception as e:
    print(sys.argv[1], 'failed', e, open('gpurun_out/b13.err').read()[-400:])
PY
done
