cd "${GRAFT_REPO_ROOT:-/root/repo}"
export PYTHONDONTWRITEBYTECODE=1
for b in 1 2 4; do for s in 1 2 3; do
  echo "== batch $b streams $s =="
  timeout 300 python bench.py --steps 8 --warmup 2 --batch $b --streams $s --no-profile --no-cpu-baseline --no-train-leg 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:r[k] for k in ('value','ms_per_ref_view')}, 'single-stream latency', r['latency']['single_stream_ms_per_ref_view'])"
done; done
