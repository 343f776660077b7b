"""Diagnostic (GPU): per-stage regulariser formats - coarse stages in the fp32-equivalent format, fine stages in the fp16 default.
    python scripts/diag_stage_mix.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import parity_cases as P
from oracle import ref_path as O
from mvsformerplusplus_amd import synth

dev = "cuda"
MIXES = {"all f16mix": ["f16mix"] * 4, "bf16x3 x1": ["bf16x3", "f16mix", "f16mix", "f16mix"], "bf16x3 x2": ["bf16x3", "bf16x3", "f16mix", "f16mix"],
         "bf16x3 x3": ["bf16x3", "bf16x3", "bf16x3", "f16mix"], "all bf16x3": ["bf16x3"] * 4}
def run(name, H, W, V, peaky=False, **inputs):
    ref = None
    for mix, precs in MIXES.items():
        head, args = P._seeded_head(dev, peaky=peaky, conv_precision="f16mix")
        for st, p in zip(head.fusions, precs):
            st.conv_precision = p
        feats, projs, dv = synth.make_cascade_inputs(H, W, V, seed=2, rot_deg=1.0, **inputs)
        if ref is None:
            sds = [{k: v.cpu() for k, v in st.state_dict().items()} for st in head.fusions]
            with torch.no_grad():
                ref = O.cascade_forward({k: v.float() for k, v in feats.items()}, projs, dv, sds, ndepths=args["ndepths"],
                                        depth_interals_ratio=args["depth_interals_ratio"], base_ch=args["base_ch"])
            ok = torch.ones(1, H, W, dtype=torch.bool)
            lo, hi = float(dv.min()) * 0.25, float(dv.max()) * 4.0
            for s in range(1, 5):
                hyp = ref["stage%d" % s]["depth_values"]
                good = (torch.isfinite(hyp) & (hyp > lo) & (hyp < hi)).all(1) & torch.isfinite(ref["stage%d" % s]["depth"])
                ok = ok & F.interpolate(good[:, None].float(), size=(H, W), mode="nearest")[:, 0].bool()
        with torch.no_grad():
            out = head({k: v.to(dev) for k, v in feats.items()}, {k: v.to(dev) for k, v in projs.items()}, dv.to(dev))
        e = ((out["refined_depth"].cpu() - ref["refined_depth"]).abs() / ref["refined_depth"].abs())[ok]
        c = (out["photometric_confidence"].cpu() - ref["photometric_confidence"]).abs()[ok]
        print("%-22s %-11s finite %.2f  depth mean %.2e median %.1e p99 %.1e   conf mean %.1e" % (name, mix, float(ok.float().mean()), float(e.mean()), float(e.median()),
              float(torch.quantile(e, 0.99)), float(c.mean())), flush=True)

c4 = P.BASELINE_CFGS["cfg4"]; inp = dict(c4["inputs"]); inp.update(depth_min=0.5, depth_interval=9.5 / (inp["numdepth"] - 1))
run("cfg4 wide range", c4["small"][0], c4["small"][1], c4["V"], **inp)
run("cfg4 0.5..3", c4["small"][0], c4["small"][1], c4["V"], **c4["inputs"])
run("midsize plain", 384, 512, 5)
run("midsize x30", 384, 512, 5, peaky=True)
