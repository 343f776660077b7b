"""Numerics study (CPU, oracle only): ONE fp16 term for the regularisers' weights.  The product default "f16x2" keeps
weights as fp16 hi + lo (22 bits) and activations as fp16 (11 bits); how far does the final depth move when the weights are fp16 too
(w_lo dropped: half the MFMAs, half the weight bytes)?  Activations rounded to fp16 in every case (the default's storage format), fp32
accumulation.  Per-layer variants: all layers / only the 32- and 64-channel layers (where w_lo costs most: 4-16 output blocks per read).
    python scripts/study_weight_precision.py [H W]"""
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
_conv3d, _convt3d, _conv2d = F.conv3d, F.conv_transpose3d, F.conv2d
NDEPTHS, RATIO = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0]
h = lambda x: x.half().float()


def state_dicts(peaky, seed=11):
    sds = []
    for i in range(4):
        net = StageNet(dict(P.ARGS), NDEPTHS[i], i)
        sd = synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), seed + i)
        if peaky:
            sd["cost_reg.prob.weight"] = sd["cost_reg.prob.weight"] * 30.0
        sds.append(sd)
    return sds


def patched(wsel):
    """activations fp16; weights of the layers `wsel(cin, cout)` picks rounded to fp16"""
    def c3(x, w, *a, **k):
        return _conv3d(h(x), h(w) if wsel(w.shape[1], w.shape[0]) else w, *a, **k)
    def ct3(x, w, *a, **k):
        return _convt3d(h(x), h(w) if wsel(w.shape[0], w.shape[1]) else w, *a, **k)
    return c3, ct3


def vis_one_term(x, w, *a, **k):
    """visibility CNN (conv2d): layers 1 - 3 run on MFMA - layer 1 (1 -> 16) with the entropy as an fp16 hi + lo pair and ONE fp16 weight term,
    layers 2 and 3 with fp16 rings (activations fp16) and ONE fp16 weight term; the 1x1 output layer is fp32 VALU code"""
    if w.shape[1] >= 8 and w.shape[-1] == 3:
        return _conv2d(h(x), h(w), *a, **k)
    if w.shape[1] == 1 and w.shape[-1] == 3:           # layer 1 on the matrix pipe: entropy as an fp16 hi + lo pair (kept exact here), weights ONE fp16 term
        return _conv2d(x, h(w), *a, **k)
    return _conv2d(x, w, *a, **k)


VIS_ONE = [False]
MODES = {"f16x2 (today: activations fp16, weights exact)": lambda ci, co: False,
         "weights fp16 on the 32/64-channel layers": lambda ci, co: min(ci, co) >= 32 or max(ci, co) >= 64,
         "weights fp16 on every layer but the 1-channel head": lambda ci, co: co > 1,
         "weights fp16 everywhere": lambda ci, co: True,
         "f16mix + visibility CNN one term": "vis",
         "round-4 default: + fp16 source windows, fp16 kept correlations": "vis+gather"}
for peaky in (False, True):
    sds = state_dicts(peaky)
    for seed in (2, 5):
        feats, projs, dv = synth.make_cascade_inputs(H, W, 5, seed=seed, rot_deg=1.0)
        run = lambda: O.cascade_forward(feats, projs, dv, sds, ndepths=NDEPTHS, depth_interals_ratio=RATIO, base_ch=P.ARGS["base_ch"])
        with torch.no_grad():
            ref = run()
            for name, wsel in MODES.items():
                gather = wsel == "vis+gather"
                vis = wsel == "vis" or gather
                fin, gc = feats, O.group_correlation
                if gather:
                    # MVS_GATHER_F16: the SOURCE views' features rounded to fp16 once; pass 1 keeps each view's group correlations as fp16
                    # on the stages with D > 4 (the last stage gathers twice and keeps nothing)
                    fin = {k: torch.cat([v[:, :1], h(v[:, 1:])], 1) for k, v in feats.items()}
                    O.group_correlation = lambda ref_f, warped, G: (h(gc(ref_f, warped, G)) if warped.shape[2] > 4 else gc(ref_f, warped, G))
                    run_g = lambda: O.cascade_forward(fin, projs, dv, sds, ndepths=NDEPTHS, depth_interals_ratio=RATIO, base_ch=P.ARGS["base_ch"])
                if vis:
                    wsel = MODES["weights fp16 on the 32/64-channel layers"]
                    F.conv2d = vis_one_term
                F.conv3d, F.conv_transpose3d = patched(wsel)
                try:
                    res = run_g() if gather else run()
                finally:
                    F.conv3d, F.conv_transpose3d, F.conv2d = _conv3d, _convt3d, _conv2d
                    O.group_correlation = gc
                errs = [rel_l1(res["stage%d" % s]["depth"], ref["stage%d" % s]["depth"]) for s in range(1, 5)]
                print("peaky=%d seed=%d  %-52s refined depth rel-L1 %.2e   stages %s   conf mean abs %.1e" % (
                    peaky, seed, name, rel_l1(res["refined_depth"], ref["refined_depth"]), " ".join("%.1e" % e for e in errs),
                    float((res["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())), flush=True)
