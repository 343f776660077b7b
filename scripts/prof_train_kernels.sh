#!/bin/bash
# which kernels the training step spends its time in (rocprofv3 kernel trace of scripts/prof_train.py, native path only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOT=$(pwd)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_train -o t -- env PROF_TRAIN_HIP_ONLY=1 python $ROOT/scripts/prof_train.py > $ROOT/gpurun_out/prof_train.log 2>&1
cd $ROOT
DB=$(find gpurun_out/prof_train -name '*.db' | head -1)
python scripts/rocpd_stats.py $DB > gpurun_out/prof_train_kernels.csv
python - <<'PY'
import csv, re
rows = list(csv.DictReader(open("gpurun_out/prof_train_kernels.csv")))
for r in rows[:28]:
    n = re.sub(r"\(.*", "", r["Name"].replace("void ", "").replace("mvs::", ""))
    print("%-78s calls %5s  total %9.2f ms  avg %9.1f us  %5s%%" % (n[:78], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
rm -rf gpurun_out/prof_train
