"""Per-stage HIP-event times of the visibility CNN at the bench workload: row-streaming single launch vs the round-1 two-launch form."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import _lib, ops
dev = torch.device("cuda:0")
head = bench.build_head(dev)
tot = {"stream": 0.0}
for s, (H, W) in enumerate(((144, 192), (288, 384), (576, 768), (1152, 1536))):
    st = head.fusions[s]
    vp = st._vis_params(dev)
    ent = torch.rand(1, 4, H, W, device=dev) * 2
    res = {}
    for impl in ("stream",):
        os.environ["MVS_VIS_IMPL"] = impl
        for _ in range(3):
            v = ops.vis_weight(ent, vp, _lib.PREC_BF16X3)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            v = ops.vis_weight(ent, vp, _lib.PREC_BF16X3)
        e1.record()
        torch.cuda.synchronize()
        res[impl] = (e0.elapsed_time(e1) / 20, v)
        tot[impl] += res[impl][0]
    os.environ["MVS_VIS_IMPL"] = "stream"
    flops = 2.0 * 4 * H * W * (9 * 16 + 9 * 16 * 16 + 9 * 16 * 8 + 8)
    print("stage %d %dx%d: %.3f ms (%.1f TFLOP/s algorithmic)" % (s + 1, H, W, res["stream"][0], flops / res["stream"][0] / 1e9))
print("totals per reference view:", {k: round(v, 3) for k, v in tot.items()})
