#!/bin/bash
# Full GPU-box visit: parity tests, smoke, bench (+ per-kernel HIP-event table), rocprofv3 kernel trace, PMC traffic pass.
# Usage: gpurun --timeout 1800 -- 'bash scripts/gpu_round.sh <tag>'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=gpurun_out; TAG=${1:-r06}
mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee $OUT/pytest_gpu.log
echo "== smoke =="
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.log
# PMC first: bench.py's roofline.traffic / whole_path.frac_pmc read profiles/pmc_traffic.json, which must describe THIS build's kernels
(cd /tmp && export TMPDIR=/tmp
echo "== PMC: HBM traffic of every kernel (separate passes) =="
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $ROOT/$OUT/pmc_$TAG/f -o f -- python $ROOT/bench.py --steps 1 --warmup 1 --views-per-step 4 --issue eager --no-shipped-leg --no-cpu-baseline --no-profile --no-train-leg > $ROOT/$OUT/pmc_f.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $ROOT/$OUT/pmc_$TAG/w -o w -- python $ROOT/bench.py --steps 1 --warmup 1 --views-per-step 4 --issue eager --no-shipped-leg --no-cpu-baseline --no-profile --no-train-leg > $ROOT/$OUT/pmc_w.log 2>&1
)
mkdir -p $OUT/profiles_$TAG
python scripts/pmc_traffic.py $OUT/pmc_$TAG $OUT/profiles_$TAG/pmc_traffic.json $OUT/profiles_$TAG/${TAG}_pmc_fetch_write_raw.json
cp $OUT/profiles_$TAG/pmc_traffic.json profiles/pmc_traffic.json
echo "== bench (driver's command) =="
timeout 1500 python bench.py --steps 20 --warmup 5 --profile-table > $OUT/bench.json 2> $OUT/bench.err
grep -v "amdgpu.ids" $OUT/bench.err | tail -45
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench.json').read().strip().splitlines()[-1])
print({k: r[k] for k in ('value', 'ms_per_step', 'ms_per_ref_view') if k in r})
for k in ('latency', 'whole_path', 'roofline', 'cpu_baseline', 'torch_rocm_composite', 'parity', 'exact_coarse_mode', 'fp32_equivalent_mode', 'uniform_f16mix_mode', 'fp16_tiles_handoff_mode', 'shipped', 'feature_emitter'):
    print(k, r.get(k))
print('gather', r.get('gather_roofline', {}).get('all_passes'))
PY
echo "== randomised cascades in the product default vs the oracle (scripts/fuzz_cascade_gpu.py 60 0) =="
timeout 600 python scripts/fuzz_cascade_gpu.py 60 0 2>&1 | grep -v amdgpu.ids > $OUT/fuzz_cascade.log; tail -3 $OUT/fuzz_cascade.log
echo "== bench, shipped regulariser mix (stage-1 transformer + PE3D) =="
timeout 900 python bench.py --steps 6 --warmup 2 --profile-table --cost-reg shipped --issue eager > $OUT/bench_shipped.json 2> $OUT/bench_shipped.err
grep -v "amdgpu.ids" $OUT/bench_shipped.err | grep -E "tr_|pos3d|softmax_regress|sum of"
python - <<'PY'
import json
r = json.loads(open('gpurun_out/bench_shipped.json').read().strip().splitlines()[-1])
print('shipped', {k: r[k] for k in ('value', 'ms_per_ref_view') if k in r}, 'parity', r.get('parity'))
PY
echo "== bench, fp32-equivalent regulariser format (--conv-precision bf16x3: split bf16 pairs, three MFMA terms) =="
timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-train-leg --no-cpu-baseline --conv-precision bf16x3 > $OUT/bench_bf16x3.json 2> $OUT/bench_bf16x3.err
python -c "
import json; r = json.loads(open('gpurun_out/bench_bf16x3.json').read().strip().splitlines()[-1]); print('bf16x3', r['value'], r['ms_per_ref_view'], r['latency']['single_stream_ms_per_ref_view'])"
echo "== bench, bf16 features in the octet-tiled hand-off layout =="
timeout 600 python bench.py --steps 6 --warmup 2 --no-profile --no-cpu-baseline --feat-layout tiled --feat-dtype bf16 > $OUT/bench_tiled_bf16.json 2>/dev/null
python -c "
import json; r = json.loads(open('gpurun_out/bench_tiled_bf16.json').read().strip().splitlines()[-1]); print('tiled bf16', r['value'], r['ms_per_ref_view'])"
echo "== bench, stages 2-4 fed by the producer-side emitter's fp16 octet tiles (no pack pass in the process; direct gather at stages 3-4) =="
timeout 600 python bench.py --steps 10 --warmup 3 --no-profile --no-train-leg --no-shipped-leg --feat-layout emitted --emit-dtype fp16 > $OUT/bench_emitted.json 2>/dev/null
python -c "
import json; r = json.loads(open('gpurun_out/bench_emitted.json').read().strip().splitlines()[-1]); print('emitted fp16 tiles', r['value'], r['ms_per_ref_view'], r.get('parity'))"
echo "== N = 2 flow of bench.py on this one-GPU box (both ranks on the device, gloo): code path check, not a measurement =="
MVS_BENCH_ONE_DEVICE=1 MVS_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --views-per-step 4 --no-profile > $OUT/bench_n2.raw 2> $OUT/bench_n2.err
grep '^{' $OUT/bench_n2.raw | tail -1 > $OUT/bench_n2.json          # gloo prints "[Gloo] Rank ..." lines on stdout: keep the JSON line only
python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench_n2.json').read().strip().splitlines()[-1])
    print('n2', {k: r[k] for k in ('value', 'n_gpus', 'ms_per_step')}, r.get('view_sharded'))
except Exception as e:
    print('bench_n2.json unreadable', e); print(open('gpurun_out/bench_n2.err').read()[-1500:])
PY
echo "== rocprofv3 kernel trace (same bench command, 5 steps) =="
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$TAG -o $TAG -- python $ROOT/bench.py --steps 5 --warmup 1 --issue eager --no-shipped-leg --no-train-leg --no-cpu-baseline --no-profile > $ROOT/$OUT/rocprof.log 2>&1
tail -2 $ROOT/$OUT/rocprof.log
# the same with ONE stream: kernels of different reference views do not overlap, so the average durations are the launches' own
# (what bench.py's HIP-event table and its `roofline` object measure); with 3 streams co-running kernels stretch each other
timeout 900 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof1_$TAG -o $TAG -- python $ROOT/bench.py --steps 5 --warmup 1 --streams 1 --issue eager --no-shipped-leg --no-train-leg --no-cpu-baseline --no-profile > $ROOT/$OUT/rocprof1.log 2>&1
cd $ROOT
DB=$(find $OUT/prof_$TAG -name '*.db' | head -1)
if [ -n "$DB" ]; then
  python scripts/rocpd_stats.py $DB > $OUT/profiles_$TAG/${TAG}_kernel_stats_whole_process.csv
  python scripts/rocpd_stats.py $DB "mvs::" > $OUT/profiles_$TAG/${TAG}_kernel_stats_3streams.csv
  DB1=$(find $OUT/prof1_$TAG -name '*.db' | head -1)
  python scripts/rocpd_stats.py $DB1 "mvs::" > $OUT/profiles_$TAG/${TAG}_kernel_stats.csv
  python scripts/group_kernel_stats.py $OUT/profiles_$TAG/${TAG}_kernel_stats.csv > $OUT/profiles_$TAG/${TAG}_kernel_stats_by_function.csv
  head -8 $OUT/profiles_$TAG/${TAG}_kernel_stats_by_function.csv
  head -14 $OUT/profiles_$TAG/${TAG}_kernel_stats.csv | cut -c1-150
else
  echo "no rocpd database under $OUT/prof_$TAG"; find $OUT/prof_$TAG | head
fi
cp $OUT/fuzz_cascade.log $OUT/profiles_$TAG/${TAG}_fuzz_cascade_default_gpu.log
cp $OUT/bench.json $OUT/profiles_$TAG/${TAG}_bench_1gpu.json
cp $OUT/bench_shipped.json $OUT/profiles_$TAG/${TAG}_bench_1gpu_shipped.json
cp $OUT/bench_n2.json $OUT/profiles_$TAG/${TAG}_bench_n2_flow_check_one_gpu_gloo.json
cp $OUT/bench_tiled_bf16.json $OUT/profiles_$TAG/${TAG}_bench_1gpu_tiled_bf16.json
cp $OUT/bench_bf16x3.json $OUT/profiles_$TAG/${TAG}_bench_1gpu_bf16x3.json
cp $OUT/bench_emitted.json $OUT/profiles_$TAG/${TAG}_bench_1gpu_emitted.json
grep -v "amdgpu.ids" $OUT/bench_bf16x3.err > $OUT/profiles_$TAG/${TAG}_bench_kernel_table_bf16x3.txt
grep -v "amdgpu.ids" $OUT/bench.err > $OUT/profiles_$TAG/${TAG}_bench_kernel_table.txt
grep -v "amdgpu.ids" $OUT/bench_shipped.err > $OUT/profiles_$TAG/${TAG}_bench_kernel_table_shipped.txt
rm -rf $OUT/prof_$TAG $OUT/prof1_$TAG $OUT/pmc_$TAG
echo "== bench, the exact-coarse branch of the default policy (--conv-precision stagemix) =="
timeout 600 python bench.py --steps 10 --warmup 3 --profile-table --no-train-leg --no-cpu-baseline --no-shipped-leg --conv-precision stagemix > $OUT/bench_stagemix.json 2> $OUT/bench_stagemix.err
python -c "
import json; r = json.loads(open('gpurun_out/bench_stagemix.json').read().strip().splitlines()[-1]); print('stagemix', r['value'], r['ms_per_ref_view'], r['latency']['single_stream_ms_per_ref_view'])"
cp $OUT/bench_stagemix.json $OUT/profiles_$TAG/${TAG}_bench_1gpu_stagemix.json
grep -v "amdgpu.ids" $OUT/bench_stagemix.err > $OUT/profiles_$TAG/${TAG}_bench_kernel_table_stagemix.txt
echo "== Track S (one StageNet at a literal D) and the U-Net layer table =="
timeout 600 python scripts/track_s_bench.py 2>&1 | grep -v amdgpu.ids | tee $OUT/profiles_$TAG/${TAG}_track_s.txt
timeout 300 python scripts/bench_unet_layers.py 2>&1 | grep -v amdgpu.ids | tee $OUT/profiles_$TAG/${TAG}_unet_layers.txt
echo "== PMC pipe utilisation per kernel (three counter passes over a short bench run) =="
bash scripts/gpu_pmc_bench.sh $TAG --no-train-leg --no-shipped-leg --issue eager --views-per-step 8 > $OUT/pmc_pipe.log 2>&1
cp $OUT/pmc_$TAG/summary.txt $OUT/profiles_$TAG/${TAG}_pmc_pipe_utilisation.txt
cp $OUT/pmc_$TAG/table.txt $OUT/profiles_$TAG/${TAG}_pmc_pipe_table_raw.txt
head -30 $OUT/profiles_$TAG/${TAG}_pmc_pipe_utilisation.txt | cut -c1-200
# the stride-1 layers run once per stage (launch i of a view = stage i + 1): the same table per stage
{ for i in 0 1 2 3; do echo "== stage $((i+1)) (launch index $i of each reference view) =="; python scripts/pmc_table_summary.py $OUT/pmc_$TAG/table.txt "ConvCfg<(16,16|32,32|64,64),3,1,1,1|vis_cnn|gl_entropy|gl_aggregate" $i | cut -c1-230; done; } > $OUT/profiles_$TAG/${TAG}_pmc_pipe_per_stage.txt
ls -la $OUT/profiles_$TAG
