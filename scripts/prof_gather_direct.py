"""HIP-event times of the fine stages' gather launches on fp16 octet tiles (the producer-side emitter's fp16 hand-off) at the bench workload
(cfg2 stage 3: C = 16, D = 8; stage 4: C = 8, D = 4).  MVS_HIP_LIB selects the build: product (direct gather, MVS_GL_DIRECT16 = 1), 0 = LDS windows,
2 / 3 = ablations (no interpolation / no loads).  Usage (GPU box): MVS_HIP_LIB=... python scripts/prof_gather_direct.py [--reps 30]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import ops, synth

ap = argparse.ArgumentParser()
ap.add_argument("--views", type=int, default=5)
ap.add_argument("--reps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda:0")
head = bench.build_head(dev)
feats, projs, dv = synth.make_cascade_inputs(1152, 1536, a.views, seed=0, device=dev)
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
    return e0.elapsed_time(e1) / reps * 1e3


print("lib:", os.environ.get("MVS_HIP_LIB") or "product")
for s in (2, 3):
    key = "stage%d" % (s + 1)
    ft = ops.pack_features(feats[key].half())
    code = ops._feat(ft)[1]
    hyp = out[key]["depth_values"].contiguous()
    hom = ops.compose_homography(projs[key])
    B, V, C, H, W = ft.shape
    vis = torch.rand(B, V - 1, H, W, device=dev)
    row = {"entropy_w16": timed(lambda: ops.warp_corr_entropy(ft, code, hom, hyp, 8, f16_window=True), a.reps),
           "aggregate_w16": timed(lambda: ops.warp_corr_aggregate(ft, code, hom, hyp, vis, 8, f16=True), a.reps)}
    if hyp.shape[1] > 4:
        row["entropy_keep"] = timed(lambda: ops.warp_corr_entropy_keep(ft, code, hom, hyp, 8), a.reps)
    print(key, "C=%d D=%d" % (C, hyp.shape[1]), "  ".join("%s %.1f us" % kv for kv in row.items()))
