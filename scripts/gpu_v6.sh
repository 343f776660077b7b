#!/bin/bash
# Round-4 visit: loader-wave convolution kernel A/B (per layer and whole bench), parity subset.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== layers, loader-wave kernels ==" | tee $OUT/r04_conv_loader_ab.txt
timeout 300 python scripts/bench_layer.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r04_conv_loader_ab.txt
echo "== layers, one-tile kernels (MVS_CONV_LOADER_OFF=1) ==" | tee -a $OUT/r04_conv_loader_ab.txt
MVS_CONV_LOADER_OFF=1 timeout 300 python scripts/bench_layer.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r04_conv_loader_ab.txt
echo "== pytest subset =="
timeout 900 python -m pytest tests -m gpu -x -q -k "golden or single_layers or f16_layers or midsize or cfg1 or cfg2" 2>&1 | tail -5 | tee $OUT/pytest_gpu_subset.log
for off in 0 1; do
  echo "== bench (MVS_CONV_LOADER_OFF=$off) ==" | tee -a $OUT/r04_conv_loader_ab.txt
  if [ $off = 1 ]; then export MVS_CONV_LOADER_OFF=1; fi
  timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-cpu-baseline --no-train-leg > $OUT/bench_l$off.json 2> $OUT/bench_l$off.err
  grep -v "amdgpu.ids" $OUT/bench_l$off.err | grep -E "conv3d|deconv|sum of" | head -24 | tee -a $OUT/r04_conv_loader_ab.txt
  python - $OUT/bench_l$off.json <<'PY' | tee -a $OUT/r04_conv_loader_ab.txt
import json, sys
r = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'latency', r['latency']['single_stream_ms_per_ref_view'], 'fam', {k: round(v['ms_per_ref_view'], 3) for k, v in r.get('families', {}).items()})
PY
done
