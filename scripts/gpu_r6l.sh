#!/bin/bash
# round 6, evidence visit: randomised differential runs on the final code - single stages (80), cascades (a second seed, 60), training path (40)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 900 python scripts/fuzz_gpu.py 80 6 2>&1 | grep -v amdgpu.ids > $OUT/r06_fuzz_stage_gpu.log; tail -2 $OUT/r06_fuzz_stage_gpu.log
timeout 900 python scripts/fuzz_cascade_gpu.py 60 1 2>&1 | grep -v amdgpu.ids > $OUT/r06_fuzz_cascade_default_gpu_seed1.log; tail -2 $OUT/r06_fuzz_cascade_default_gpu_seed1.log
timeout 900 python scripts/fuzz_train_gpu.py 40 2>&1 | grep -v amdgpu.ids > $OUT/r06_fuzz_train_gpu.log; tail -2 $OUT/r06_fuzz_train_gpu.log
