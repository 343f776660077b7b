"""Numerics study (CPU, oracle only), round 5: on SURVEY section 8d's literal 0.5 .. 10 range (cfg4 / cfg5: ill-conditioned around the
pixels whose inverse-depth window crosses zero) WHICH storage form of the coarse stages' gather carries the error the "stagemix" policy
still shows (1.0e-3 / 1.3e-3) - the fp16 source windows or the fp16 kept correlations?  Fine stages (D <= 8) always in the fp16 default.
    python scripts/study_wide_range_gather.py [cfg4|cfg5]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F
import parity_cases as P
from oracle import ref_path as O
from mvsformerplusplus_amd import synth
from mvsformerplusplus_amd.cost_volume import StageNet

_conv3d, _convt3d, _conv2d, _gc, _sf = F.conv3d, F.conv_transpose3d, F.conv2d, O.group_correlation, O.stage_forward
NDEPTHS, RATIO = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0]
h = lambda x: x.half().float()
CUR = {"D": 0}
one_term = lambda ci, co: min(ci, co) >= 32 or max(ci, co) >= 64


def install(reg_f16, win_f16, corr_f16, vis_f16=None):
    def sf(features, proj_matrices, depth_values, *a, **k):
        CUR["D"] = depth_values.shape[1]
        if win_f16(CUR["D"]):
            features = torch.cat([features[:, :1], h(features[:, 1:])], 1)
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
        return h(r) if corr_f16(CUR["D"]) and warped.shape[2] > 4 else r
    O.stage_forward, F.conv3d, F.conv_transpose3d, F.conv2d, O.group_correlation = sf, c3, ct3, c2, gc


def restore():
    O.stage_forward, F.conv3d, F.conv_transpose3d, F.conv2d, O.group_correlation = _sf, _conv3d, _convt3d, _conv2d, _gc


fine = lambda D: D <= 8
yes = lambda D: True
MODES = [("fp16 default on every stage", yes, yes, yes),
         ("coarse exact (reg + gather), fine fp16", fine, fine, fine),
         ("coarse: exact reg, fp16 windows only", fine, yes, fine),
         ("coarse: exact reg, fp16 kept corr only", fine, fine, yes),
         ("coarse: exact reg, fp16 windows + kept corr (stagemix)", fine, yes, yes),
         ("stage 1 exact all; stage 2 exact reg + fp16 gather", fine, lambda D: D <= 16, lambda D: D <= 16),
         ("coarse: fp16 reg, exact gather", yes, fine, fine),
         ("coarse exact reg + gather, fp16 (one-term) vis CNN", fine, fine, fine, yes)]

for name in (sys.argv[1:] or ["cfg4"]):
    c = P.BASELINE_CFGS[name]
    inputs = dict(c["inputs"]); nd = inputs["numdepth"]
    inputs.update(depth_min=0.5, depth_interval=9.5 / (nd - 1))
    H, W, V = c["small"][0], c["small"][1], c["V"]
    sds = []
    for i in range(4):
        net = StageNet(dict(P.ARGS), NDEPTHS[i], i)
        sds.append(synth.seeded_state_dict(synth.state_dict_manifest(net.state_dict()), 11 + i))
    feats, projs, dv = synth.make_cascade_inputs(H, W, V, seed=2, rot_deg=1.0, **inputs)
    feats = {k: v.float() for k, v in feats.items()}
    run = lambda: O.cascade_forward(feats, projs, dv, sds, ndepths=NDEPTHS, depth_interals_ratio=RATIO, base_ch=P.ARGS["base_ch"])
    with torch.no_grad():
        ref = run()
        ok = torch.ones(1, H, W, dtype=torch.bool)
        lo, hi = float(dv.min()) * 0.25, float(dv.max()) * 4.0
        for s in range(1, 5):
            hyp = ref["stage%d" % s]["depth_values"]
            good = (torch.isfinite(hyp) & (hyp > lo) & (hyp < hi)).all(1) & torch.isfinite(ref["stage%d" % s]["depth"])
            ok = ok & F.interpolate(good[:, None].float(), size=(H, W), mode="nearest")[:, 0].bool()
        print("%s wide range: finite fraction %.2f" % (name, float(ok.float().mean())), flush=True)
        for mname, rf, wf, cf, *vf in MODES:
            install(rf, wf, cf, *vf)
            try:
                res = run()
            finally:
                restore()
            e = ((res["refined_depth"] - ref["refined_depth"]).abs() / ref["refined_depth"].abs())[ok]
            print("%s  %-58s mean %.2e  median %.1e  p99 %.1e" % (name, mname, float(e.mean()), float(e.median()),
                                                                  float(torch.quantile(e, 0.99))), flush=True)
