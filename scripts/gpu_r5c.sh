#!/bin/bash
# Round 5, visit 3: where does the exact-coarse-stage policy cost its 0.17 ms?  Per-stage wall times per policy + one-stream rocprofv3 kernel
# totals per reference view for the default policy and for the uniform f16mix format.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
ROOT=$(pwd); OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
timeout 300 python scripts/diag_stage_time.py 2>&1 | grep -v amdgpu.ids
cd /tmp && export TMPDIR=/tmp
for pol in stagemix f16mix; do
  timeout 600 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof_$pol -o $pol -- python $ROOT/bench.py --steps 2 --warmup 1 --views-per-step 8 --streams 1 --issue eager \
      --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg --conv-precision $pol > $ROOT/$OUT/rocprof_$pol.log 2>&1
  DB=$(find $ROOT/$OUT/prof_$pol -name '*.db' | head -1)
  python $ROOT/scripts/rocpd_stats.py $DB "mvs::" > $ROOT/$OUT/kernel_stats_$pol.csv
  rm -rf $ROOT/$OUT/prof_$pol
  echo "== $pol: mvs:: kernels, one stream (calls, total us, avg us) =="
  python - $ROOT/$OUT/kernel_stats_$pol.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total %.1f us over all views" % (tot / 1e3))
for r in rows[:46]:
    print("%6d %10.1f %8.1f  %s" % (int(r["Calls"]), int(r["TotalDurationNs"]) / 1e3, float(r["AverageNs"]) / 1e3, r["Name"][:150]))
PY
done
