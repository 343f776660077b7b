#!/bin/bash
# Round 5, visit 6: stage 4 (D = 4) with kept fp16 correlations + streamed pass 2 (KEEP_MIN_DEPTH = 1) against the second gather (default, 5)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python -m pytest tests -m gpu -x -q -k "gather_variants or gather_windows" 2>&1 | tail -2
for args in "" "--keep-min-depth 1" "" "--keep-min-depth 1"; do
  timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg --no-shipped-leg $args > $OUT/ab.json 2> $OUT/ab.err
  grep -E "gl_|corr_aggregate|sum of" $OUT/ab.err
  python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('%-24s' % '$args', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms; single', round(r['latency']['single_stream_ms_per_ref_view'],3), 'gather', round(r['families']['gather']['ms_per_ref_view'],3), 'vs bf16x3 rel', r['fp32_equivalent_mode']['default_vs_this_refined_depth_rel_l1'])"
done
