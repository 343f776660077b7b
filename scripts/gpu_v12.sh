#!/bin/bash
# Short GPU-box visit (last session of round 4, < 8 GPU-minutes left): the ABI-v10 additions on the real library -
# shape-generic convolution kernel, base_ch != 8 stages, range-function branches - then smoke and a timing of the generic U-Net.
# Usage: gpurun --timeout 300 -- 'bash scripts/gpu_v12.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== pytest -m gpu: the new cases =="
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider \
  -k "generic or other_groups or range_variants or small_fns or stage_golden or regnet_golden" 2>&1 | tail -6 | tee $OUT/pytest_gpu_v12.log
echo "== smoke =="
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke_v12.log
echo "== generic U-Net timing (CostRegNet3D(G, G) at cfg2 stage-3 / stage-4 volumes, CostRegNet(G, G) at stage 2) =="
timeout 120 python - <<'PY' 2>&1 | tee $OUT/generic_timing_v12.txt
import time, torch
from mvsformerplusplus_amd import module as M, synth
dev = torch.device("cuda", 0)
for name, make, shape in (("CostRegNet3D(4,4)  stage 4 [4,1152,1536]", lambda: M.CostRegNet3D(4, 4), (1, 4, 4, 1152, 1536)),
                          ("CostRegNet3D(16,16) stage 3 [8,576,768]", lambda: M.CostRegNet3D(16, 16), (1, 16, 8, 576, 768)),
                          ("CostRegNet(4,4)    stage 2 [16,288,384]", lambda: M.CostRegNet(4, 4), (1, 4, 16, 288, 384)),
                          ("CostRegNet3D(8,8) tuned f16mix, stage 4 (for scale)", lambda: M.CostRegNet3D(8, 8), (1, 8, 4, 1152, 1536))):
    net = make()
    net.load_state_dict(synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), 3))
    net = net.eval().to(dev)
    x = torch.randn(*shape, device=dev)
    with torch.no_grad():
        net(x); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
    print("%-55s %8.2f ms per forward (NCDHW in, logits out)" % (name, (time.perf_counter() - t0) / 3 * 1e3))
PY
