"""Per-stage depth error of the HIP cascade vs the oracle with 'peaky' logits (prob weights x30), both precisions."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import parity_cases as P
from conftest import rel_l1
from mvsformerplusplus_amd import synth
from oracle import ref_path as O

dev = "cuda"
for peaky in (False, True):
    head, args = P._seeded_head(dev, peaky=peaky)
    feats, projs, dv = synth.make_cascade_inputs(384, 512, 5, seed=2, rot_deg=1.0)
    sds = [{k: v.cpu() for k, v in st.state_dict().items()} for st in head.fusions]
    with torch.no_grad():
        ref = O.cascade_forward(feats, projs, dv, sds, ndepths=args["ndepths"], depth_interals_ratio=args["depth_interals_ratio"], base_ch=args["base_ch"])
    for prec in ("fp32", "bf16x3"):
        for st in head.fusions:
            st.conv_precision = prec
        with torch.no_grad():
            out = head({k: v.to(dev) for k, v in feats.items()}, {k: v.to(dev) for k, v in projs.items()}, dv.to(dev))
        errs = [rel_l1(out["stage%d" % s]["depth"].cpu(), ref["stage%d" % s]["depth"]) for s in range(1, 5)]
        mx = [float(((out["stage%d" % s]["depth"].cpu() - ref["stage%d" % s]["depth"]).abs() / ref["stage%d" % s]["depth"]).max()) for s in range(1, 5)]
        conf = float((out["photometric_confidence"].cpu() - ref["photometric_confidence"]).abs().max())
        maxprob = float(ref["stage1"]["photometric_confidence"].mean())
        print("peaky=%s %-6s stage rel-L1 %s  max-rel %s  conf max-abs %.3g  (mean max-prob stage1 %.2f)" % (peaky, prec, ["%.2e" % e for e in errs], ["%.1e" % e for e in mx], conf, maxprob))
