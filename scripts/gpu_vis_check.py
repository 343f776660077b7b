import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import parity_cases as P
from conftest import load_golden, golden_weights
from mvsformerplusplus_amd import _lib, ops
from oracle import ref_path as O
fx = load_golden("f2_stage_s1.npz"); sd = golden_weights(fx)
ref = O.stage_forward(fx["features"], fx["proj"], fx["hyp"], 5.0, sd, G=8, return_intermediates=True)
net = P.make_stage(fx, fx["hyp"].shape[1], 1, "cuda")
ent = ref["entropy"].squeeze(2).cuda().contiguous()
for prec in ("fp32", "bf16x3"):
    net.conv_precision = prec
    for rep in range(3):
        vis = ops.vis_weight(ent, net._vis_params(ent.device), _lib.PRECISIONS[prec]).cpu()
        d = (vis - ref["vis_weight"].squeeze(2)).abs()
        print(prec, rep, "max", float(d.max()), "mean", float(d.mean()), "nan", int(torch.isnan(vis).sum()), "argmax", [int(i) for i in torch.nonzero(d == d.max())[0]])
