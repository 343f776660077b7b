#!/bin/bash
# round 6, visit e: full GPU suite on the round-6 default ("auto"), streams sweep under it, Track S with final_stage, one full default bench line
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
B="--steps 8 --warmup 2 --no-profile --no-cpu-baseline --no-train-leg --no-shipped-leg"
for s in 3 4 5 4 3 5; do
  echo "== streams $s =="
  timeout 300 python bench.py $B --streams $s 2>gpurun_out/r6e_err.txt | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:round(r[k],3) for k in ('value','ms_per_ref_view')}, 'single-stream latency', round(r['latency']['single_stream_ms_per_ref_view'],3))" || tail -3 gpurun_out/r6e_err.txt
done
echo "== track S =="
timeout 600 python scripts/track_s_bench.py 2>&1 | grep -v amdgpu.ids
echo "== full default bench =="
timeout 900 python bench.py > gpurun_out/r6e_bench.json 2> gpurun_out/r6e_bench.err; tail -3 gpurun_out/r6e_bench.err
python - <<'P'
import json
r = json.loads(open('gpurun_out/r6e_bench.json').read().strip().splitlines()[-1])
print('headline', round(r['value'], 1), r['config'].get('precision_policy'), 'parity', r.get('parity'))
for k in ('exact_coarse_mode', 'fp32_equivalent_mode', 'fp16_tiles_handoff_mode', 'shipped', 'torch_rocm_composite', 'cpu_baseline'):
    v = r.get(k, {})
    print(k, {kk: v.get(kk) for kk in ('value', 'ms_per_ref_view', 'error', 'parity', 'hip_path_vs_this_refined_depth_rel_l1', 'peak_memory_gb', 'sample')})
print('roofline', {k: r['roofline'][k] for k in ('kernel', 'achieved', 'frac', 'traffic', 'avg_launch_ms')})
print('families', {k: round(v['ms_per_ref_view'], 3) for k, v in r['families'].items()})
P
