#!/bin/bash
# Tile-size re-check of the MFMA convolutions on the round-4 formats (f16mix): product library vs one variant library per tile knob
# (libmvs_hip_<tag>.so next to it, built with build.build(extra_flags=[-DMVS_T...], out=...)).  Whole-path throughput only.
# Usage: gpurun --timeout 300 -- 'bash scripts/gpu_v14.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
: > $OUT/tiles_ab_v14.txt
for lib in mvsformerplusplus_amd/csrc/libmvs_hip.so mvsformerplusplus_amd/csrc/libmvs_hip_*.so mvsformerplusplus_amd/csrc/libmvs_hip.so; do
  [ -f "$lib" ] || continue
  tag=$(basename $lib .so)
  MVS_HIP_LIB=$PWD/$lib timeout 100 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg > $OUT/b14.json 2> $OUT/b14.err
  python - "$tag" <<'PY' | tee -a $OUT/tiles_ab_v14.txt
import json, sys
try:
    r = json.loads(open('gpurun_out/b14.json').read().strip().splitlines()[-1])
    print("%-28s %7.1f ref-views/s  %6.3f ms/view  single-stream %6.3f ms" % (sys.argv[1], r['value'], r['ms_per_ref_view'], r['latency']['single_stream_ms_per_ref_view']))
except Exception as e:
    print(sys.argv[1], 'failed', e, open('gpurun_out/b14.err').read()[-400:])
PY
done
