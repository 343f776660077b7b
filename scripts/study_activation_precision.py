"""Numerics study (CPU, oracle only): how far does the final depth move when every activation tensor the 3-D regularisers CONSUME
(cost volume, U-Net intermediates incl. the skip sums, the `prob` head's input) is rounded to a narrower storage format?
fp32 weights and fp32 accumulation throughout: this isolates the storage precision of the activations, i.e. whether the bf16x3
contraction's `w . x_lo` term and the lo halves of the split activation format could be dropped.
    python scripts/study_activation_precision.py [H W]
Cases: the 4-stage cascade on synthetic rigs, plain and 'peaky' (prob weights x30: the stress set of tests/parity_cases.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import parity_cases as P
from conftest import rel_l1
from oracle import ref_path as O
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cost_volume import StageNet

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 320)
ROUND = {"fp32": lambda x: x,
         "bf16 (8 bits)": lambda x: x.bfloat16().float(),
         "fp16 (11 bits)": lambda x: x.half().float(),
         "bf16 hi + bf16 lo (16 bits, today)": lambda x: (lambda h: h + (x - h).bfloat16().float())(x.bfloat16().float())}
_conv3d, _convt3d = F.conv3d, F.conv_transpose3d
NDEPTHS, RATIO = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0]


def state_dicts(peaky, seed=11):
    sds = []
    for i in range(4):
        net = StageNet(dict(P.ARGS), NDEPTHS[i], i)
        sd = synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), seed + i)
        if peaky:
            sd["cost_reg.prob.weight"] = sd["cost_reg.prob.weight"] * 30.0
        sds.append(sd)
    return sds


for peaky in (False, True):
    sds = state_dicts(peaky)
    for seed in (2, 5):
        feats, projs, dv = synth.make_cascade_inputs(H, W, 5, seed=seed, rot_deg=1.0)
        res = {}
        for name, rnd in ROUND.items():
            F.conv3d = lambda x, *a, _r=rnd, **k: _conv3d(_r(x), *a, **k)
            F.conv_transpose3d = lambda x, *a, _r=rnd, **k: _convt3d(_r(x), *a, **k)
            try:
                with torch.no_grad():
                    res[name] = O.cascade_forward(feats, projs, dv, sds, ndepths=NDEPTHS, depth_interals_ratio=RATIO, base_ch=P.ARGS["base_ch"])
            finally:
                F.conv3d, F.conv_transpose3d = _conv3d, _convt3d
        ref = res["fp32"]
        for name in ROUND:
            if name == "fp32":
                continue
            errs = [rel_l1(res[name]["stage%d" % s]["depth"], ref["stage%d" % s]["depth"]) for s in range(1, 5)]
            print("peaky=%d seed=%d  %-36s refined depth rel-L1 %.2e   stages %s   conf mean abs %.1e" % (
                peaky, seed, name, rel_l1(res[name]["refined_depth"], ref["refined_depth"]), " ".join("%.1e" % e for e in errs),
                float((res[name]["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())), flush=True)
