"""Diagnostic (GPU): error distribution of the cascade on the degenerate 0.5 .. 10 range (case_cascade_vs_oracle_finite) per format.
    python scripts/diag_wide_range.py [cfg4|cfg5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import parity_cases as P
from oracle import ref_path as O
from mvsformerplusplus_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
c = P.BASELINE_CFGS[name]
inputs = dict(c["inputs"]); nd = inputs["numdepth"]
inputs.update(depth_min=0.5, depth_interval=9.5 / (nd - 1))
H, W, V = c["small"][0], c["small"][1], c["V"]
dev = "cuda"
ref = None
MODES = (("bf16x3", "auto"), ("stagemix", "auto"), ("stagemix", False), ("f16mix", "auto"))
if len(sys.argv) > 2 and sys.argv[2] == "all":
    MODES += (("f16x2", False), ("f16x2", "auto"), ("f16mix", False), ("f16", "auto"))
for prec, keep in MODES:
    head, args = P._seeded_head(dev, conv_precision=prec)
    for st in head.fusions:
        st.keep_correlations = keep
    feats, projs, dv = synth.make_cascade_inputs(H, W, V, seed=2, rot_deg=1.0, **inputs)
    if ref is None:
        sds = [{k: v.cpu() for k, v in st.state_dict().items()} for st in head.fusions]
        with torch.no_grad():
            ref = O.cascade_forward({k: v.float() for k, v in feats.items()}, projs, dv, sds, ndepths=args["ndepths"],
                                    depth_interals_ratio=args["depth_interals_ratio"], base_ch=args["base_ch"])
        ok = torch.ones(1, H, W, dtype=torch.bool)
        lo, hi = float(dv.min()) * 0.25, float(dv.max()) * 4.0
        margins = []
        for s in range(1, 5):
            hyp = ref["stage%d" % s]["depth_values"]
            good = (torch.isfinite(hyp) & (hyp > lo) & (hyp < hi)).all(1) & torch.isfinite(ref["stage%d" % s]["depth"])
            ok = ok & F.interpolate(good[:, None].float(), size=(H, W), mode="nearest")[:, 0].bool()
        print("finite fraction %.2f" % float(ok.float().mean()))
    with torch.no_grad():
        out = head({k: v.to(dev) for k, v in feats.items()}, {k: v.to(dev) for k, v in projs.items()}, dv.to(dev))
    d, r = out["refined_depth"].cpu(), ref["refined_depth"]
    e = ((d - r).abs() / r.abs())[ok]
    q = torch.quantile(e, torch.tensor([0.5, 0.9, 0.99, 0.999]))
    per_stage = []
    for s in range(1, 5):
        ds_, rs_ = out["stage%d" % s]["depth"].cpu(), ref["stage%d" % s]["depth"]
        oks = F.interpolate(ok[:, None].float(), size=ds_.shape[-2:], mode="nearest")[:, 0].bool()
        per_stage.append(float(((ds_ - rs_).abs() / rs_.abs())[oks].mean()))
    print("%-7s keep=%-5s mean %.2e  median %.1e  p90 %.1e  p99 %.1e  p99.9 %.1e  frac>1e-2 %.4f  max %.2e  stages %s" % (
        prec, keep, float(e.mean()), *[float(x) for x in q], float((e > 1e-2).float().mean()), float(e.max()), " ".join("%.1e" % x for x in per_stage)), flush=True)
