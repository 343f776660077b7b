#!/bin/bash
# Round 5: the cascade fed by the producer-side emitter (bench.py --feat-layout emitted): throughput + parity, and a rocprofv3 kernel list of the
# whole process showing conv2d3x3_tiles_kernel and NO pack_features_kernel (VERDICT r4 item 8's done-criterion)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 600 python bench.py --steps 10 --warmup 3 --no-train-leg --feat-layout emitted > $OUT/bench_emitted.json 2> $OUT/bench_emitted.err
python -c "
import json; r = json.loads(open('gpurun_out/bench_emitted.json').read().strip().splitlines()[-1]); print('emitted', round(r['value'],1), 'ref-views/s', round(r['ms_per_ref_view'],3), 'ms; parity', r.get('parity'), r['config']['features'])"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_emit -o emit -- python $ROOT/bench.py --steps 2 --warmup 1 --views-per-step 8 --issue eager --no-shipped-leg --no-train-leg --no-cpu-baseline --no-profile --feat-layout emitted > $ROOT/$OUT/rocprof_emit.log 2>&1
cd $ROOT
DB=$(find $OUT/prof_emit -name '*.db' | head -1)
python scripts/rocpd_stats.py $DB "mvs::" > $OUT/kernel_stats_emitted.csv
rm -rf $OUT/prof_emit
echo "pack_features launches: $(grep -c pack_features $OUT/kernel_stats_emitted.csv); emitter launches:"; grep conv2d3x3_tiles $OUT/kernel_stats_emitted.csv | cut -c1-160
