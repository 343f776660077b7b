"""Numerics study (CPU, oracle only): WHICH cascade stage's fp16 noise ends up in the refined depth?  The coarse stages (D = 32, 16:
CostRegNet) schedule the hypotheses of the fine ones, so their noise is amplified; they are also the cheap ones.  Modes: the fp16
default on every stage / exact arithmetic on the coarse stages with the fp16 default on the fine ones / the same with the coarse
stages' GATHER still in its fp16 storage forms (fp16 source windows, per-view correlations kept as fp16) - the form that keeps the
streaming pass 2.   python scripts/study_stage_mix.py [H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import parity_cases as P
from conftest import rel_l1
from oracle import ref_path as O
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cost_volume import StageNet

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 320)
_conv3d, _convt3d, _conv2d, _gc, _sf = F.conv3d, F.conv_transpose3d, F.conv2d, O.group_correlation, O.stage_forward
NDEPTHS, RATIO = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0]
h = lambda x: x.half().float()
CUR = {"D": 0}
one_term = lambda ci, co: min(ci, co) >= 32 or max(ci, co) >= 64


def state_dicts(peaky, seed=11):
    sds = []
    for i in range(4):
        net = StageNet(dict(P.ARGS), NDEPTHS[i], i)
        sd = synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), seed + i)
        if peaky:
            sd["cost_reg.prob.weight"] = sd["cost_reg.prob.weight"] * 30.0
        sds.append(sd)
    return sds


def install(reg_f16, gather_f16, vis_f16=None):
    """reg_f16(D) / gather_f16(D): does the stage with D hypotheses run the fp16 regulariser (+ visibility CNN) format / the fp16 gather forms?"""
    def sf(features, proj_matrices, depth_values, *a, **k):
        CUR["D"] = depth_values.shape[1]
        if gather_f16(CUR["D"]):
            features = torch.cat([features[:, :1], h(features[:, 1:])], 1)       # fp16 source windows
        return _sf(features, proj_matrices, depth_values, *a, **k)
    def c3(x, w, *a, **k):
        if not reg_f16(CUR["D"]):
            return _conv3d(x, w, *a, **k)
        return _conv3d(h(x), h(w) if one_term(w.shape[1], w.shape[0]) else w, *a, **k)
    def ct3(x, w, *a, **k):
        if not reg_f16(CUR["D"]):
            return _convt3d(x, w, *a, **k)
        return _convt3d(h(x), h(w) if one_term(w.shape[0], w.shape[1]) else w, *a, **k)
    vis_f16 = vis_f16 or reg_f16
    def c2(x, w, *a, **k):
        if not vis_f16(CUR["D"]) or w.shape[-1] != 3:
            return _conv2d(x, w, *a, **k)
        return _conv2d(h(x) if w.shape[1] >= 8 else x, h(w), *a, **k)
    def gc(ref_f, warped, G):
        r = _gc(ref_f, warped, G)
        return h(r) if gather_f16(CUR["D"]) and warped.shape[2] > 4 else r       # kept correlations (D > 4)
    O.stage_forward, F.conv3d, F.conv_transpose3d, F.conv2d, O.group_correlation = sf, c3, ct3, c2, gc


def restore():
    O.stage_forward, F.conv3d, F.conv_transpose3d, F.conv2d, O.group_correlation = _sf, _conv3d, _convt3d, _conv2d, _gc


MODES = [("fp16 default on every stage (round 4 so far)", lambda D: True, lambda D: True),
         ("coarse stages (D > 8) exact, fine stages fp16", lambda D: D <= 8, lambda D: D <= 8),
         ("coarse stages: exact regulariser, fp16 gather forms", lambda D: D <= 8, lambda D: True),
         ("stage 1 only exact regulariser, fp16 gather forms", lambda D: D <= 16, lambda D: True),
         ("coarse: exact U-Net, fp16 gather AND fp16 vis CNN", lambda D: D <= 8, lambda D: True, lambda D: True)]
for peaky in (False, True):
    sds = state_dicts(peaky)
    for seed in (2, 5):
        feats, projs, dv = synth.make_cascade_inputs(H, W, 5, seed=seed, rot_deg=1.0)
        run = lambda: O.cascade_forward(feats, projs, dv, sds, ndepths=NDEPTHS, depth_interals_ratio=RATIO, base_ch=P.ARGS["base_ch"])
        with torch.no_grad():
            ref = run()
            for name, rf, gf, *vf in MODES:
                install(rf, gf, *vf)
                try:
                    res = run()
                finally:
                    restore()
                errs = [rel_l1(res["stage%d" % s]["depth"], ref["stage%d" % s]["depth"]) for s in range(1, 5)]
                print("peaky=%d seed=%d  %-52s refined depth rel-L1 %.2e   stages %s   conf mean abs %.1e" % (
                    peaky, seed, name, rel_l1(res["refined_depth"], ref["refined_depth"]), " ".join("%.1e" % e for e in errs),
                    float((res["photometric_confidence"] - ref["photometric_confidence"]).abs().mean())), flush=True)
