#!/bin/bash
# Is the 1.5 ms path still clear of the host?  Eager issue (~67 launches per view from Python) vs hipGraph replay (bench.py --graph), and the
# host's issue time per view measured directly.  Usage: gpurun --timeout 240 -- 'bash scripts/gpu_v15.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
: > $OUT/graph_ab_v15.txt
nproc | sed 's/^/host threads: /' | tee -a $OUT/graph_ab_v15.txt
for cfg in "--streams 3" "--streams 3 --graph" "--streams 4 --graph" "--streams 6 --graph" "--streams 3 --graph --input-sets 2" "--streams 3"; do
  timeout 120 python bench.py --steps 10 --warmup 3 --no-profile --no-cpu-baseline --no-train-leg $cfg > $OUT/b15.json 2> $OUT/b15.err
  python - "$cfg" <<'PY' | tee -a $OUT/graph_ab_v15.txt
import json, sys
try:
    r = json.loads(open('gpurun_out/b15.json').read().strip().splitlines()[-1])
    print("%-40s %7.1f ref-views/s  %6.3f ms/view  single-stream %6.3f ms" % (sys.argv[1], r['value'], r['ms_per_ref_view'], r['latency']['single_stream_ms_per_ref_view']))
except Exception as e:
    print(sys.argv[1], 'failed', e, open('gpurun_out/b15.err').read()[-600:])
PY
done
timeout 60 python - <<'PY' | tee -a $OUT/graph_ab_v15.txt
import time, torch, bench
from mvsformerplusplus_amd import synth
dev = torch.device("cuda", 0)
head = bench.build_head(dev)
f, p, d = synth.make_cascade_inputs(1152, 1536, 5, seed=1, device=dev)
with torch.no_grad():
    for _ in range(3):
        head(f, p, d, tmp=bench.TMP)
    torch.cuda.synchronize()
    # host issue time: launches queue up behind the device (the stream is deep enough for a few views), the clock stops before the sync
    t0 = time.perf_counter()
    for _ in range(6):
        head(f, p, d, tmp=bench.TMP)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
print("host issue time per reference view (6 views back to back, one stream): %.3f ms; device-complete: %.3f ms per view" % ((t1 - t0) / 6 * 1e3, (t2 - t0) / 6 * 1e3))
PY
