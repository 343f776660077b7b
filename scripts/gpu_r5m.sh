#!/bin/bash
# round 5, visit m: direct gather from fp16 octet tiles (MVS_GL_DIRECT16) against the LDS-window form on the same tiles - tests, then A/B
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 400 python -m pytest tests -m gpu -x -q -k "gather_variants or gather_windows or feature_heads or stage_lowp" 2>&1 | tail -2
B="--steps 10 --warmup 3 --no-profile --no-train-leg --no-shipped-leg --feat-layout emitted --emit-dtype fp16"
run() {   # tag, lib
    MVS_HIP_LIB="$2" timeout 300 python bench.py $B > gpurun_out/ab_$1.json 2>gpurun_out/ab_$1.err || tail -5 gpurun_out/ab_$1.err
    python -c "
import json,sys; r = json.loads(open('gpurun_out/ab_$1.json').read().strip().splitlines()[-1]); print('$1', round(r['value'],1), 'ref-views/s', 'parity', (r.get('parity') or {}).get('refined_depth_rel_l1_vs_oracle'), '| single-stream ms', round(r['latency']['single_stream_ms_per_ref_view'],3))"
}
L=mvsformerplusplus_amd/csrc
run direct1 ""
B="$B --no-cpu-baseline"
run lds1 $L/libmvs_hip_nodirect.so
run direct2 ""
run lds2 $L/libmvs_hip_nodirect.so
# per-kernel view: one stream, eager, both forms
cd /tmp && export TMPDIR=/tmp
for t in direct lds; do
    lib=""; [ $t = lds ] && lib=$GRAFT_REPO_ROOT/$L/libmvs_hip_nodirect.so
    MVS_HIP_LIB="$lib" timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$t -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --streams 1 --issue eager --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg --feat-layout emitted --emit-dtype fp16 > /dev/null 2>$GRAFT_REPO_ROOT/gpurun_out/prof_$t.err
    DB=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_$t -name '*.db' | head -1)
    echo "== $t: $DB"
    if [ -n "$DB" ]; then
        python $GRAFT_REPO_ROOT/scripts/rocpd_stats.py $DB "mvs::" > $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$t.csv
        grep -i "gl_\|corr_agg" $GRAFT_REPO_ROOT/gpurun_out/kernel_stats_$t.csv | cut -c1-170
        rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$t
    fi
done
