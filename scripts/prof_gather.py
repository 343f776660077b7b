"""Per-stage HIP-event times of the two gather passes at the bench workload (cfg2): workgroup-window LDS kernels (default) vs
direct (round 1, MVS_GATHER_IMPL=direct) kernels.  MVS_HIP_LIB selects a variant build.
Usage (GPU box): python scripts/prof_gather.py [--views 5] [--reps 20] [--impls lds,direct]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--height", type=int, default=1152)
ap.add_argument("--width", type=int, default=1536)
ap.add_argument("--feat-dtype", default="fp32")
ap.add_argument("--impls", default="lds,direct")
a = ap.parse_args()
dev = torch.device("cuda:0")
head = bench.build_head(dev)
fdt = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp16": torch.float16}[a.feat_dtype]
feats, projs, dv = synth.make_cascade_inputs(a.height, a.width, a.views, seed=0, device=dev, feat_dtype=fdt)
with torch.no_grad():
    out = head(feats, projs, dv, tmp=bench.TMP)
torch.cuda.synchronize()


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


tot = {}
for s in range(4):
    key = "stage%d" % (s + 1)
    f, code = ops._feat(feats[key])
    hyp = out[key]["depth_values"].contiguous()
    hom = ops.compose_homography(projs[key])
    B, V, C, H, W = f.shape
    D = hyp.shape[1]
    vis = torch.rand(B, V - 1, H, W, device=dev)
    row = {}
    res = {}
    impls = a.impls.split(",")
    for impl in impls:
        os.environ["MVS_GATHER_IMPL"] = impl
        row[impl + "_entropy"] = timed(lambda: ops.warp_corr_entropy(f, code, hom, hyp, 8), a.reps)
        row[impl + "_aggregate"] = timed(lambda: ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8), a.reps)
        res[impl] = (ops.warp_corr_entropy(f, code, hom, hyp, 8), ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8)[0])
    os.environ.pop("MVS_GATHER_IMPL", None)
    de = max(float((res[impls[0]][0] - res[i][0]).abs().max()) for i in impls)
    dvv = max(float((res[impls[0]][1] - res[i][1]).abs().max()) for i in impls)
    esz = f.element_size()
    alg_e = B * (V * C * H * W * esz + D * H * W * 4 + (V - 1) * H * W * 4)
    alg_a = B * (V * C * H * W * esz + D * H * W * 4 + (V - 1) * H * W * 4 + 8 * D * H * W * 4)
    print("stage %d C=%d D=%d %dx%d: entropy %s | aggregate %s | max|d entropy| %.2e max|d vol| %.2e"
          % (s + 1, C, D, H, W, "  ".join("%s %.3f ms (%.0f GB/s)" % (i, row[i + "_entropy"], alg_e / row[i + "_entropy"] / 1e6) for i in impls),
             "  ".join("%s %.3f ms (%.0f GB/s)" % (i, row[i + "_aggregate"], alg_a / row[i + "_aggregate"] / 1e6) for i in impls), de, dvv))
    for k, v in row.items():
        tot[k] = tot.get(k, 0.0) + v
print("totals per reference view:", {k: round(v, 3) for k, v in tot.items()})
