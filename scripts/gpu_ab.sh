#!/bin/bash
# GPU visit: parity subset, then bench (per-kernel table) for the product library and for every variant library libmvs_hip_*.so
# found next to it (built with mvsformerplusplus_amd.build.build(extra_flags=[...], out=...)).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu (subset: ${PYTEST_K:-all}) =="
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q ${PYTEST_K:+-k "$PYTEST_K"} 2>&1 | tail -4 | tee $OUT/pytest_gpu_ab.log
for lib in mvsformerplusplus_amd/csrc/libmvs_hip.so mvsformerplusplus_amd/csrc/libmvs_hip_*.so; do
    [ -f "$lib" ] || continue
    tag=$(basename $lib .so)
    echo "== bench: $tag =="
    MVS_HIP_LIB=$PWD/$lib timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg ${BENCH_ARGS:-} > $OUT/bench_$tag.json 2> $OUT/bench_$tag.err
    grep -v "amdgpu.ids" $OUT/bench_$tag.err | grep -E "${TABLE_GREP:-conv|deconv|sum of|kernel  }" | head -${TABLE_ROWS:-30}
    python - "$OUT/bench_$tag.json" <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r.get('latency', {}).get('single_stream_ms_per_ref_view'), 'parity', r.get('parity', {}).get('refined_depth_rel_l1_vs_oracle'))
except Exception as e:
    print('bench json unreadable', e)
PY
done
