"""Where the time of the LDS-staged gather passes goes: the passes of the round-6 default policy ("auto" -> fp16 gather forms: keeping pass 1 at stages 1-3,
plain pass 1 + second gather at stage 4) at cfg2's shapes, HIP-event timing, with variant libraries built with -DMVS_GL_ABL=n (gather_lds.h: parts of the
unit disabled - the results are then wrong, only the time counts).  Usage (GPU box): MVS_HIP_LIB=<variant .so> python scripts/gather_ablate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from mvsformerplusplus_amd import ops, synth

dev = torch.device("cuda:0")
head = bench.build_head(dev)
feats, projs, dv = synth.make_cascade_inputs(1152, 1536, 5, seed=0, device=dev)
HYP = "gpurun_out/gather_ablate_hyps.pt"            # the stages' hypotheses come from the PRODUCT library's cascade (run it first), not from an ablated one
if os.environ.get("MVS_HIP_LIB"):
    hyps = torch.load(HYP, map_location=dev)
else:
    with torch.no_grad():
        out = head(feats, projs, dv, tmp=bench.TMP)
    torch.cuda.synchronize()
    hyps = {"stage%d" % (s + 1): out["stage%d" % (s + 1)]["depth_values"].contiguous() for s in range(4)}
    torch.save({k: v.cpu() for k, v in hyps.items()}, HYP)


def timed(fn, reps=20):
    best = 1e9
    for _ in range(3):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    return best


row = []
for s in range(4):
    key = "stage%d" % (s + 1)
    f, code = ops._feat(feats[key])
    hyp = hyps[key].to(dev).contiguous()
    hom = ops.compose_homography(projs[key])
    B, V, C, H, W = f.shape
    vis = torch.rand(B, V - 1, H, W, device=dev)
    if s < 3:
        row.append("st%d keep %6.1f" % (s + 1, timed(lambda: ops.warp_corr_entropy_keep(f, code, hom, hyp, 8))))
    else:
        row.append("st4 pass1 %6.1f" % timed(lambda: ops.warp_corr_entropy(f, code, hom, hyp, 8, f16_window=True)))
        row.append("st4 pass2 %6.1f" % timed(lambda: ops.warp_corr_aggregate(f, code, hom, hyp, vis, 8, normalise=True, f16=True)))
print(os.path.basename(os.environ.get("MVS_HIP_LIB", "") or "product"), " | ".join(row), "us", flush=True)
