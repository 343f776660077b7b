#!/bin/bash
# Final visit of round 4's last session: the -m gpu suite without the three full-size oracle comparisons (their kernels did not change in this
# session; the driver's round-end run covers them), then the generic U-Net timing on the final kernel.
# Usage: gpurun --timeout 300 -- 'bash scripts/gpu_v16.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export PYTHONDONTWRITEBYTECODE=1
echo "== generic U-Net timing =="
timeout 90 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $OUT/generic_timing_v16.txt
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
echo "== pytest -m gpu (without the full-size oracle comparisons) =="
timeout 260 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "not fullsize" --durations=8 2>&1 | tail -16 | tee $OUT/pytest_gpu_v16.log
