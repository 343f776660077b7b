#!/bin/bash
# Round 5, visit 2: new tests (fused head + schedule, producer emitter, Track S D = 192 past the 2 GB ceiling), bench with the folded
# launches, stream-count and KEEP_EXACT_MIN_DEPTH A/B in graph-replay mode.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu (new / touched cases) =="
timeout 900 python -m pytest tests -m gpu -x -q -k "track_s or fused_small or feature_heads or regnet_golden or single_layers or f16_layers or cascade_golden or gather_variants or cfg1 or hip_graph" 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
echo "== bench (driver's command) =="
timeout 900 python bench.py --steps 20 --warmup 5 --profile-table > $OUT/bench.json 2> $OUT/bench.err
grep -v "amdgpu.ids" $OUT/bench.err | tail -44
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_step', 'ms_per_ref_view') if k in r})
    for k in ('latency', 'roofline', 'parity', 'fp32_equivalent_mode', 'shipped', 'feature_emitter'):
        print(k, r.get(k))
    print('families', {k: (v['ms_per_ref_view'], v.get('launches_per_ref_view')) for k, v in r.get('families', {}).items()})
except Exception as e:
    print('bench.json unreadable', e)
PY
for args in "--streams 3" "--streams 4" "--streams 5" "--streams 6" "--streams 3 --keep-exact-min-depth 32" "--streams 3 --keep-exact-min-depth 1000" "--streams 3 --conv-precision f16mix" "--streams 3"; do
  timeout 600 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg $args > $OUT/ab.json 2>/dev/null
  python -c "
import json; r = json.loads(open('gpurun_out/ab.json').read().strip().splitlines()[-1]); print('%-48s' % '$args', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms; single', round(r['latency']['single_stream_ms_per_ref_view'],3))"
done
