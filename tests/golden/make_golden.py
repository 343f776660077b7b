#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

``/root/reference`` is a read-only mount that exists only in the build container; it never travels
to the GPU box.  What is committed is DATA: inputs, weights (reference state-dict naming) and the
reference's outputs as ``.npz`` - no reference source.  Fixture list = SURVEY.md §8c F1-F6.
"""
import io
import json
import os
import sys

import numpy as np

REF = os.environ.get("MVS_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)
sys.dont_write_bytecode = True

import torch  # noqa: E402

from models.cost_volume import StageNet  # noqa: E402  (reference)
from models.module import (CostRegNet, CostRegNet2D, CostRegNet3D, conf_regression, depth_regression, init_inverse_range,  # noqa: E402
                           init_range, schedule_inverse_range, schedule_range)
from models.warping import homo_warping_3D_with_mask  # noqa: E402

from mvsformerplusplus_amd import synth  # noqa: E402

torch.set_num_threads(8)
ARGS = {"base_ch": [8, 8, 8, 8], "depth_type": ["ce"] * 4, "fusion_type": "cnn", "cost_reg_type": ["Normal"] * 4}


def npz(name, **arrs):
    out = {}
    for k, v in arrs.items():
        if torch.is_tensor(v):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print("%-36s %8.1f KB  %s" % (name, os.path.getsize(path) / 1024, sorted(out)[:6]))


def seed_weights(module, seed, prefix="w."):
    """Load deterministic weights (synth.seeded_state_dict) into a reference module; return the manifest
    arrays to store in the fixture (keys + shapes + seed), not the weights themselves."""
    man = synth.state_dict_manifest(module.state_dict())
    module.load_state_dict(synth.seeded_state_dict(man, seed), strict=True)
    keys = list(man)
    return {prefix + "keys": np.array(keys), prefix + "shapes": np.array([json.dumps(list(man[k])) for k in keys]),
            prefix + "seed": np.int64(seed)}


def rand_cams(B, V, H, W, seed, rot_deg, baseline, scale=1.0):
    cams = synth.make_cameras(V, H, W, baseline=baseline, rot_deg=rot_deg, seed=seed, batch=B)
    cams[:, :, 1, :2, :] *= scale
    return cams


def compose(p):
    o = p[:, 0].clone()
    o[:, :3, :4] = torch.matmul(p[:, 1, :3, :3], p[:, 0, :3, :4])
    return o


@torch.no_grad()
def f1_warp():
    for tag, (B, C, H, W, D), rot, base in (("a", (1, 8, 16, 20, 4), 3.0, 40.0), ("b", (2, 16, 12, 16, 6), 8.0, 120.0)):
        g = torch.Generator().manual_seed(11 + B)
        cams = rand_cams(B, 2, H * 8, W * 8, 5 + B, rot, base, scale=1 / 8)
        # make a slice of hypotheses sit behind the source camera / far out of frame
        cams[:, 1, 0, 2, 3] = -300.0 if tag == "b" else 0.0
        src_fea = torch.randn(B, C, H, W, generator=g)
        ref_p, src_p = compose(cams[:, 0]), compose(cams[:, 1])
        dv2 = torch.linspace(200.0, 900.0, D)[None].repeat(B, 1).contiguous()
        dv4 = (dv2[:, :, None, None] * (1 + 0.1 * torch.rand(B, D, H, W, generator=g))).contiguous()
        w2, m2 = homo_warping_3D_with_mask(src_fea, src_p, ref_p, dv2)
        w4, m4 = homo_warping_3D_with_mask(src_fea, src_p, ref_p, dv4)
        npz("f1_warp_%s.npz" % tag, src_fea=src_fea, src_proj=src_p, ref_proj=ref_p, dv2=dv2, dv4=dv4,
            warped2=w2, mask2=m2, warped4=w4, mask4=m4)


def stage_inputs(C, D, H, W, V, seed, dmin=425.0, dmax=935.0, down=1):
    cams = rand_cams(1, V, H * down, W * down, seed, 2.0, 30.0, scale=1.0 / down)
    feats = synth.make_features(cams, C, H, W, dmin=dmin + 60, dmax=dmax - 60, seed=seed, noise=0.1)
    g = torch.Generator().manual_seed(seed)
    base = 1.0 / torch.linspace(1.0 / dmax, 1.0 / dmin, D)            # far -> near like init_inverse_range
    hyp = (base[None, :, None, None] * (1 + 0.02 * torch.rand(1, D, H, W, generator=g))).contiguous()
    return feats, cams, hyp


@torch.no_grad()
def f2_stage():
    for tag, stage_idx, C, D in (("s1", 1, 32, 16), ("s3", 3, 8, 4)):
        torch.manual_seed(20 + stage_idx)
        net = StageNet(dict(ARGS), D, stage_idx).eval()
        wman = seed_weights(net, 200 + stage_idx)
        feats, cams, hyp = stage_inputs(C, D, 32, 40, 3, 30 + stage_idx, down=2 ** (3 - stage_idx))
        cap = {}
        h = net.cost_reg.register_forward_hook(lambda m, i, o: cap.__setitem__("vol", i[0].detach().clone()))
        out = net(feats, cams, hyp, tmp=5.0 if stage_idx < 3 else 1.0)
        h.remove()
        npz("f2_stage_%s.npz" % tag, features=feats, proj=cams, hyp=hyp, tmp=np.float32(5.0 if stage_idx < 3 else 1.0),
            stage_idx=np.int32(stage_idx), volume_mean=cap["vol"], depth=out["depth"], prob_volume=out["prob_volume"],
            photometric_confidence=out["photometric_confidence"], prob_volume_pre=out["prob_volume_pre"],
            **wman)


@torch.no_grad()
def f14_stage_bd_hypotheses():
    """StageNet.forward with depth_values of shape [B, D] (fronto-parallel planes shared by every pixel; the reference's warp broadcasts them,
    warping.py:91, and depth_regression views them as [B, D, 1, 1], module.py:650-652) - the call form of a plain (non-cascade) MVSNet."""
    stage_idx, C, D = 3, 8, 8
    torch.manual_seed(61)
    net = StageNet(dict(ARGS), D, stage_idx).eval()
    wman = seed_weights(net, 261)
    feats, cams, _ = stage_inputs(C, D, 32, 40, 3, 61, down=1)
    hyp = (1.0 / torch.linspace(1.0 / 935.0, 1.0 / 425.0, D))[None].contiguous()            # [1, D]
    out = net(feats, cams, hyp, tmp=1.0)
    npz("f14_stage_bd_hyp.npz", features=feats, proj=cams, hyp=hyp, depth=out["depth"], prob_volume=out["prob_volume"],
        photometric_confidence=out["photometric_confidence"], prob_volume_pre=out["prob_volume_pre"], **wman)


@torch.no_grad()
def f15_stage_other_groups():
    """StageNet with base_ch != 8 (cost_volume.py:29-49: in_channels = base_ch, CostRegNet(G, G) / CostRegNet3D(G, G), widths 2G / 4G / 8G):
    G = 4 < C on a CostRegNet stage (D = 16) and on a CostRegNet3D stage (D = 4), G = C = 16 (the per-channel branch :83-85) on D = 4."""
    for tag, stage_idx, C, G, D, H, W in (("g4_s1", 1, 16, 4, 16, 32, 40), ("g4_s3", 3, 8, 4, 4, 32, 40), ("g16_s2", 2, 16, 16, 4, 16, 24)):
        torch.manual_seed(70 + G + stage_idx)
        args = dict(ARGS, base_ch=[G] * 4)
        net = StageNet(args, D, stage_idx).eval()
        wman = seed_weights(net, 270 + G + stage_idx)
        feats, cams, hyp = stage_inputs(C, D, H, W, 3, 70 + G + stage_idx, down=2 ** (3 - stage_idx))
        cap = {}
        h = net.cost_reg.register_forward_hook(lambda m, i, o: cap.__setitem__("vol", i[0].detach().clone()))
        tmp = 5.0 if stage_idx < 3 else 1.0
        out = net(feats, cams, hyp, tmp=tmp)
        h.remove()
        npz("f15_stage_%s.npz" % tag, features=feats, proj=cams, hyp=hyp, tmp=np.float32(tmp), stage_idx=np.int32(stage_idx),
            base_ch=np.int32(G), volume_mean=cap["vol"], depth=out["depth"], prob_volume=out["prob_volume"],
            photometric_confidence=out["photometric_confidence"], prob_volume_pre=out["prob_volume_pre"], **wman)


@torch.no_grad()
def f16_regnet_inner():
    """Regularisers whose in_channels differ from base_channels (the 1x1x1 `inner` convolution, module.py:385-388 / 481-484), other
    base widths, CostRegNet(last_layer=False) (features out, :406-408) and CostRegNet3D(log_var=True) (two `prob` channels, :486)."""
    g = torch.Generator().manual_seed(16)
    cases = (("crn_4_8", lambda: CostRegNet(4, 8), (1, 4, 8, 16, 16)),
             ("crn_12_4", lambda: CostRegNet(12, 4), (2, 12, 8, 8, 16)),
             ("crn_8_8_nolast", lambda: CostRegNet(8, 8, last_layer=False), (1, 8, 8, 8, 16)),
             ("crn3d_12_8", lambda: CostRegNet3D(12, 8), (1, 12, 4, 16, 24)),
             ("crn3d_6_6", lambda: CostRegNet3D(6, 6), (1, 6, 3, 8, 16)),
             ("crn3d_8_8_logvar", lambda: CostRegNet3D(8, 8, log_var=True), (1, 8, 4, 8, 16)))
    arrs = {}
    for i, (tag, make, shape) in enumerate(cases):
        torch.manual_seed(160 + i)
        net = make().eval()
        arrs.update(seed_weights(net, 160 + i, prefix=tag + ".w."))
        x = torch.randn(*shape, generator=g)
        arrs[tag + ".x"] = x
        arrs[tag + ".y"] = net.forward_once(x)
    npz("f16_regnet_inner.npz", **arrs)


@torch.no_grad()
def f19_position_encoding():
    """get_position_3d(normalize=True / False) and PositionEncoding3D as a tensor (position_encoding.py:138-189) on a small frustum."""
    from models.position_encoding import PositionEncoding3D, get_position_3d
    g = torch.Generator().manual_seed(19)
    B, D, H, W = 2, 3, 6, 8
    K = torch.tensor([[[40.0, 0.0, 4.0], [0.0, 42.0, 3.0], [0.0, 0.0, 1.0]]]).repeat(B, 1, 1)
    K[1, 0, 0] = 36.0
    hyp = (torch.linspace(900, 450, D)[None, :, None, None] * (1 + 0.05 * torch.rand(B, D, H, W, generator=g))).contiguous()
    pos, hmin, hmax, wmin, wmax = get_position_3d(B, H, W, K, hyp, 425.0, 935.0, None, None, None, None, normalize=True)
    raw = get_position_3d(B, H, W, K, hyp, 425.0, 935.0, None, None, None, None, normalize=False)[0]
    arrs = {"K": K, "hyp": hyp, "position3d": pos, "ranges": torch.stack([hmin, hmax, wmin, wmax]), "position3d_raw": raw}
    for C, rescale in ((8, 4.0), (6, 2.5)):
        arrs["pe_c%d" % C] = PositionEncoding3D(pos, C, rescale=rescale)
    npz("f19_position_encoding.npz", **arrs)


@torch.no_grad()
def f18_costregnet2d():
    """CostRegNet2D (module.py:411-450; dead code in the reference, kept for class-level API parity): base 8 and base 4."""
    g = torch.Generator().manual_seed(18)
    arrs = {}
    for i, (tag, base, shape) in enumerate((("b8", 8, (1, 8, 4, 16, 24)), ("b4", 4, (2, 4, 3, 8, 16)))):
        torch.manual_seed(180 + i)
        net = CostRegNet2D(base, base).eval()
        arrs.update(seed_weights(net, 180 + i, prefix=tag + ".w."))
        x = torch.randn(*shape, generator=g)
        arrs[tag + ".x"] = x
        arrs[tag + ".y"] = net(x)
    npz("f18_costregnet2d.npz", **arrs)


def pin_weights():
    """tests/golden/weights_sha256.json: SHA-256 of every weight set the fixtures regenerate from (manifest, seed) - checked on every load
    (tests/conftest.py golden_weights)."""
    import glob
    import hashlib
    sums = {}
    for path in sorted(glob.glob(os.path.join(HERE, "*.npz"))):
        z = np.load(path, allow_pickle=False)
        for key in z.files:
            if not key.endswith("keys"):
                continue
            prefix = key[:-4]
            keys = [str(k) for k in z[prefix + "keys"].tolist()]
            shapes = [tuple(json.loads(str(x))) for x in z[prefix + "shapes"].tolist()]
            sd = synth.seeded_state_dict(dict(zip(keys, shapes)), int(z[prefix + "seed"]))
            h = hashlib.sha256()
            for k in sorted(sd):
                h.update(k.encode())
                h.update(sd[k].contiguous().numpy().tobytes())
            sums[os.path.basename(path) + ":" + prefix] = h.hexdigest()
    old = json.load(open(os.path.join(HERE, "weights_sha256.json")))
    for k, v in old.items():
        assert sums.get(k, v) == v, "weights of %s changed" % k
    json.dump(sums, open(os.path.join(HERE, "weights_sha256.json"), "w"), indent=1, sort_keys=True)
    print("pinned %d weight sets (%d new)" % (len(sums), len(set(sums) - set(old))))


@torch.no_grad()
def f3_regnets():
    g = torch.Generator().manual_seed(3)
    torch.manual_seed(3)
    net = CostRegNet(8, 8).eval()
    wman = seed_weights(net, 31)
    x = torch.randn(1, 8, 16, 16, 24, generator=g)
    npz("f3_costregnet.npz", x=x, y=net.forward_once(x), **wman)
    for D in (4, 8):
        net = CostRegNet3D(8, 8).eval()
        wman = seed_weights(net, 32 + D)
        x = torch.randn(1, 8, D, 16, 24, generator=g)
        npz("f3_costregnet3d_d%d.npz" % D, x=x, y=net.forward_once(x), **wman)


@torch.no_grad()
def f4_cascade():
    """4-stage cascade from features, 64x128, V=4, all-"Normal" (logic of DINOv2_mvsformer_model.py:120-179)."""
    import torch.nn.functional as F
    H, W, V = 64, 128, 4
    ndepths, ratios, tmp = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0], [5.0, 5.0, 5.0, 1.0]
    feats, projs, dv = synth.make_cascade_inputs(H, W, V, seed=4, baseline=30.0, rot_deg=1.0)
    torch.manual_seed(40)
    nets = [StageNet(dict(ARGS), ndepths[i], i).eval() for i in range(4)]
    arrs = {"depth_values": dv}
    for i, n in enumerate(nets):
        arrs.update(seed_weights(n, 40 + i, "w%d." % (i + 1)))
    prob_maps = torch.zeros(1, H, W)
    st = None
    for s in range(4):
        f, p = feats["stage%d" % (s + 1)], projs["stage%d" % (s + 1)]
        h, w = f.shape[-2:]
        if s == 0:
            hyp = init_inverse_range(dv, ndepths[s], dv.device, dv.dtype, h, w)
        else:
            hyp = schedule_inverse_range(st["depth"].detach(), st["depth_values"], ndepths[s], ratios[s], h, w)
        st = nets[s](f, p, hyp, tmp=tmp[s])
        conf = st["photometric_confidence"]
        if conf.shape[1] != H or conf.shape[2] != W:
            conf = F.interpolate(conf.unsqueeze(1), [H, W], mode="nearest").squeeze(1)
        prob_maps += conf
        arrs["features%d" % (s + 1)] = f
        arrs["proj%d" % (s + 1)] = p
        arrs["hyp%d" % (s + 1)] = hyp
        arrs["depth%d" % (s + 1)] = st["depth"]
        arrs["conf%d" % (s + 1)] = st["photometric_confidence"]
    arrs["refined_depth"] = st["depth"]
    arrs["photometric_confidence"] = prob_maps / 4
    # features dominate the size: keep them fp16-representable so the .npz compresses well but stays exact
    npz("f4_cascade.npz", **arrs)


@torch.no_grad()
def f5_small_fns():
    g = torch.Generator().manual_seed(5)
    arrs = {}
    for D, n in ((32, 4), (16, 3), (8, 2)):
        p = torch.softmax(3 * torch.randn(2, D, 6, 7, generator=g), 1)
        dv = torch.sort(torch.rand(2, D, 6, 7, generator=g) * 500 + 400, dim=1, descending=True)[0]
        arrs["p%d" % D] = p
        arrs["dv%d" % D] = dv
        arrs["dreg%d" % D] = depth_regression(p, dv)
        arrs["conf%d_n%d" % (D, n)] = conf_regression(p, n=n)
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None].repeat(2, 1)
    dv[1] = dv[1] * 1.3
    arrs["depth_values"] = dv
    arrs["init_range"] = init_range(dv, 8, dv.device, dv.dtype, 5, 6)
    arrs["init_inverse_range"] = init_inverse_range(dv, 8, dv.device, dv.dtype, 5, 6)
    prev_hyp = init_inverse_range(dv, 8, dv.device, dv.dtype, 5, 6) * (1 + 0.01 * torch.rand(2, 8, 5, 6, generator=g))
    prev_depth = prev_hyp[:, 3] * (1 + 0.02 * torch.rand(2, 5, 6, generator=g))
    arrs["prev_hyp"] = prev_hyp
    arrs["prev_depth"] = prev_depth
    arrs["schedule_inverse_range"] = schedule_inverse_range(prev_depth, prev_hyp, 4, 2.67, 10, 12)
    itv = (dv[:, 1] - dv[:, 0]) * 1.5
    arrs["schedule_range_itv"] = itv
    arrs["schedule_range"] = schedule_range(prev_depth, 4, itv, 10, 12)
    npz("f5_small_fns.npz", **arrs)


@torch.no_grad()
def f17_range_variants():
    """The branches of the range functions no shipped config takes: per-pixel initial ranges [B,H,W,N] (module.py:683-688, 698-703),
    schedule_inverse_range(shift=True) (:712-715; the DTU-like depths put most pixels below the 0.002 floor, so the branch fires) and
    per-pixel depth intervals [B,H/2,W/2] in schedule_range (:731-732)."""
    g = torch.Generator().manual_seed(17)
    arrs = {}
    lo = 400 + 100 * torch.rand(2, 5, 6, 1, generator=g)
    px = torch.cat([lo, lo + 200, lo + 400 + 100 * torch.rand(2, 5, 6, 1, generator=g)], -1).contiguous()      # [B,H,W,3], first < last
    arrs["pixel_ranges"] = px
    arrs["init_range_pixel"] = init_range(px, 8, px.device, px.dtype, 5, 6)
    arrs["init_inverse_range_pixel"] = init_inverse_range(px, 8, px.device, px.dtype, 5, 6)
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None].repeat(2, 1)
    prev_hyp = init_inverse_range(dv, 8, dv.device, dv.dtype, 5, 6) * (1 + 0.01 * torch.rand(2, 8, 5, 6, generator=g))
    plane = torch.randint(0, 8, (2, 1, 5, 6), generator=g)                  # a surface anywhere in the range: near planes stay above the floor
    prev_depth = torch.gather(prev_hyp, 1, plane)[:, 0] * (1 + 0.02 * torch.rand(2, 5, 6, generator=g))
    arrs["prev_hyp"] = prev_hyp
    arrs["prev_depth"] = prev_depth
    arrs["schedule_inverse_range_shift"] = schedule_inverse_range(prev_depth, prev_hyp, 4, 1.0, 10, 12, shift=True)
    inv_max = 1 / prev_depth - 1.0 * (1. / prev_hyp[:, 2] - 1. / prev_hyp[:, 1])
    assert 0.2 < float((inv_max < 0.002).float().mean()) < 0.95, "the fixture must exercise both sides of the shift floor"
    itv = 2.65 * (1 + torch.rand(2, 5, 6, generator=g))
    arrs["schedule_range_itv_pixel"] = itv
    arrs["schedule_range_pixel"] = schedule_range(prev_depth, 4, itv, 10, 12)
    npz("f17_range_variants.npz", **arrs)


@torch.no_grad()
def f6_train_mode():
    torch.manual_seed(6)
    net = StageNet(dict(ARGS), 8, 2)
    wman = seed_weights(net, 6)
    feats, cams, hyp = stage_inputs(16, 8, 16, 24, 3, 36, down=2)
    # train-mode BN would use batch statistics; the 'ce' argmax branch (cost_volume.py:109-112) is what this
    # fixture pins, so BN layers are put in eval mode while self.training stays True on StageNet.
    net.train()
    for m in net.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            m.eval()
    out = net(feats, cams, hyp, tmp=5.0)
    npz("f6_stage_train_ce.npz", features=feats, proj=cams, hyp=hyp, depth=out["depth"],
        photometric_confidence=out["photometric_confidence"], prob_volume=out["prob_volume"], **wman)

    # 'reg' depth type, eval (cost_volume.py:119-128) for D = 8 (conf_regression n=2)
    args = dict(ARGS)
    args["depth_type"] = ["reg"] * 4
    net = StageNet(args, 8, 2).eval()
    wman = seed_weights(net, 7)
    out = net(feats, cams, hyp, tmp=1.0)
    npz("f6_stage_reg.npz", features=feats, proj=cams, hyp=hyp, depth=out["depth"],
        photometric_confidence=out["photometric_confidence"], **wman)


def f12_train_backward():
    """Full train mode (BatchNorm batch statistics, checkpointed regulariser) forward + backward of the reference StageNet:
    gradients w.r.t. the features and every parameter, and the running statistics after the step."""
    for tag, stage_idx, C, D in (("s3", 3, 8, 4), ("s1", 1, 32, 16)):
        torch.manual_seed(120 + stage_idx)
        net = StageNet(dict(ARGS), D, stage_idx).train()
        wman = seed_weights(net, 1200 + stage_idx)
        parts = [stage_inputs(C, D, 16, 24, 3, 130 + stage_idx + 7 * b, down=2 ** (3 - stage_idx)) for b in range(2)]
        feats = torch.cat([p[0] for p in parts]).requires_grad_(True)
        cams = torch.cat([p[1] for p in parts])
        hyp = torch.cat([p[2] for p in parts])
        R = torch.randn(2, D, 16, 24, generator=torch.Generator().manual_seed(5 + stage_idx))
        out = net(feats, cams, hyp, tmp=1.0)
        loss = (out["prob_volume"] * R).sum() + 0.05 * out["prob_volume_pre"].pow(2).mean()
        loss.backward()
        grads = {"g." + k: v.grad for k, v in net.named_parameters()}
        assert all(g is not None for g in grads.values())
        stats = {"stat." + k: v for k, v in net.state_dict().items() if "running_" in k}
        npz("f12_train_backward_%s.npz" % tag, features=feats, proj=cams, hyp=hyp, R=R, stage_idx=np.int32(stage_idx), loss=loss,
            prob_volume_pre=out["prob_volume_pre"], depth=out["depth"], photometric_confidence=out["photometric_confidence"],
            g_features=feats.grad, **grads, **stats, **wman)


TRANSFORMER_CFG = {"base_channel": 8, "mid_channel": 64, "num_heads": 4, "down_rate": [2, 4, 4], "mlp_ratio": 4, "layer_num": 6,
                   "drop": 0.0, "attn_drop": 0.0, "position_encoding": True, "attention_type": "FLASH2",
                   "softmax_scale": "entropy_invariance", "train_avg_length": 12185, "use_pe_proj": True}   # config/mvsformer++.json:94-113
SHIPPED_ARGS = dict(ARGS, cost_reg_type=["PureTransformerCostReg", "Normal", "Normal", "Normal"], use_pe3d=True,
                    transformer_config=[TRANSFORMER_CFG])


@torch.no_grad()
def f7_transformer():
    """PureTransformerCostReg alone (module.py:602-646) with the Frustoconical PE of position_encoding.py:138-189."""
    from models.module import PureTransformerCostReg
    from models.position_encoding import get_position_3d
    g = torch.Generator().manual_seed(7)
    net = PureTransformerCostReg(8, **dict(TRANSFORMER_CFG)).eval()
    wman = seed_weights(net, 70)
    B, D, H, W = 1, 8, 16, 24
    x = torch.randn(B, 8, D, H, W, generator=g)
    K = torch.tensor([[[361.5, 0.0, 12.0], [0.0, 361.5, 8.0], [0.0, 0.0, 1.0]]])
    hyp = ((1.0 / torch.linspace(1 / 900.0, 1 / 430.0, D))[None, :, None, None] * (1 + 0.02 * torch.rand(1, D, H, W, generator=g))).contiguous()
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None]
    pos, hmin, hmax, wmin, wmax = get_position_3d(B, H, W, K, hyp, depth_min=dv.min(), depth_max=dv.max(), height_min=None,
                                                   height_max=None, width_min=None, width_max=None, normalize=True)
    npz("f7_transformer.npz", x=x, K=K, hyp=hyp, depth_values=dv, position3d=pos, pe_range=torch.stack([hmin, hmax, wmin, wmax]),
        y=net(x, pos), y_nope=net(x, None), cfg=np.array(json.dumps(TRANSFORMER_CFG)), **wman)


@torch.no_grad()
def f8_stage_transformer():
    """StageNet stage_idx 0 with the shipped transformer regulariser (cost_volume.py:41-43), V=3, 16x24, D=32."""
    from models.position_encoding import get_position_3d
    args = json.loads(json.dumps(SHIPPED_ARGS))
    net = StageNet(args, 32, 0).eval()
    wman = seed_weights(net, 80)
    feats, cams, hyp = stage_inputs(64, 32, 16, 24, 3, 38, down=8)
    feats = feats.half().float()                         # stored as fp16 (exactly representable) to keep the fixture small
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None]
    pos = get_position_3d(1, 16, 24, cams[:, 0, 1, :3, :3], hyp, depth_min=dv.min(), depth_max=dv.max(), height_min=None,
                          height_max=None, width_min=None, width_max=None, normalize=True)[0]
    out = net(feats, cams, hyp, tmp=5.0, position3d=pos)
    npz("f8_stage_transformer.npz", features=feats.half(), proj=cams, hyp=hyp, position3d=pos, depth=out["depth"],
        prob_volume=out["prob_volume"], photometric_confidence=out["photometric_confidence"], prob_volume_pre=out["prob_volume_pre"],
        cfg=np.array(json.dumps(TRANSFORMER_CFG)), **wman)


def f13_train_backward_transformer():
    """The shipped stage 1 (transformer regulariser + Frustoconical PE) in train mode: loss, logits and the gradients of the features
    and of every parameter."""
    from models.position_encoding import get_position_3d
    args = json.loads(json.dumps(SHIPPED_ARGS))
    torch.manual_seed(13)
    net = StageNet(args, 32, 0).train()
    wman = seed_weights(net, 1300)
    feats, cams, hyp = stage_inputs(64, 32, 16, 24, 3, 138, down=8)
    feats = feats.half().float().requires_grad_(True)
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None]
    with torch.no_grad():
        pos = get_position_3d(1, 16, 24, cams[:, 0, 1, :3, :3], hyp, depth_min=dv.min(), depth_max=dv.max(), height_min=None,
                              height_max=None, width_min=None, width_max=None, normalize=True)[0]
    R = torch.randn(1, 32, 16, 24, generator=torch.Generator().manual_seed(14))
    out = net(feats, cams, hyp, tmp=1.0, position3d=pos)
    loss = (out["prob_volume"] * R).sum() + 0.05 * out["prob_volume_pre"].pow(2).mean()
    loss.backward()
    grads = {"g." + k: v.grad for k, v in net.named_parameters()}
    assert all(g is not None for g in grads.values())
    npz("f13_train_backward_transformer.npz", features=feats.detach().half(), proj=cams, hyp=hyp, position3d=pos, R=R, loss=loss,
        prob_volume_pre=out["prob_volume_pre"], g_features=feats.grad, cfg=np.array(json.dumps(TRANSFORMER_CFG)), **grads, **wman)


@torch.no_grad()
def f9_cascade_shipped():
    """The f4 cascade inputs through the SHIPPED regulariser mix (stage-1 transformer + PE3D), driver logic of
    DINOv2_mvsformer_model.py:120-179.  Inputs are the ones stored in f4_cascade.npz."""
    import torch.nn.functional as F
    from models.position_encoding import get_position_3d
    H, W, V = 64, 128, 4
    ndepths, ratios, tmp = [32, 16, 8, 4], [4.0, 2.67, 1.5, 1.0], [5.0, 5.0, 5.0, 1.0]
    feats, projs, dv = synth.make_cascade_inputs(H, W, V, seed=4, baseline=30.0, rot_deg=1.0)
    nets = [StageNet(json.loads(json.dumps(SHIPPED_ARGS)), ndepths[i], i).eval() for i in range(4)]
    arrs = {"cfg": np.array(json.dumps(TRANSFORMER_CFG))}
    for i, n in enumerate(nets):
        arrs.update(seed_weights(n, 90 + i, "w%d." % (i + 1)))
    prob_maps = torch.zeros(1, H, W)
    st = None
    rng = [None] * 4
    for s in range(4):
        f, p = feats["stage%d" % (s + 1)], projs["stage%d" % (s + 1)]
        h, w = f.shape[-2:]
        hyp = init_inverse_range(dv, ndepths[s], dv.device, dv.dtype, h, w) if s == 0 else \
            schedule_inverse_range(st["depth"].detach(), st["depth_values"], ndepths[s], ratios[s], h, w)
        pos = None
        if SHIPPED_ARGS["cost_reg_type"][s] != "Normal":
            pos, *rng = get_position_3d(1, h, w, p[:, 0, 1, :3, :3], hyp, depth_min=dv.min(), depth_max=dv.max(), height_min=rng[0],
                                        height_max=rng[1], width_min=rng[2], width_max=rng[3], normalize=True)
        st = nets[s](f, p, hyp, tmp=tmp[s], position3d=pos)
        conf = st["photometric_confidence"]
        if conf.shape[1] != H or conf.shape[2] != W:
            conf = F.interpolate(conf.unsqueeze(1), [H, W], mode="nearest").squeeze(1)
        prob_maps += conf
        arrs["depth%d" % (s + 1)] = st["depth"]
        arrs["conf%d" % (s + 1)] = st["photometric_confidence"]
    arrs["refined_depth"] = st["depth"]
    arrs["photometric_confidence"] = prob_maps / 4
    npz("f9_cascade_shipped.npz", **arrs)


def fusion_inputs(n_src=4, h=48, w=64, seed=10):
    """Depth maps of one tilted plane seen from translated cameras (closed form), plus noise / outliers / holes so that every
    mask of the filters is exercised; confidences in 0..1."""
    g = torch.Generator().manual_seed(seed)
    cams = synth.make_cameras(n_src + 1, h, w, baseline=30.0, rot_deg=0.0, seed=seed)          # [1,V,2,4,4], R = I
    a, b, z0 = 0.15, -0.1, 600.0
    ys, xs = torch.meshgrid(torch.arange(h, dtype=torch.float32) + 0.5, torch.arange(w, dtype=torch.float32) + 0.5, indexing="ij")
    depths = []
    for v in range(n_src + 1):
        K, E = cams[0, v, 1, :3, :3], cams[0, v, 0]
        C = -E[:3, 3]
        rx, ry = (xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1]
        depths.append((z0 + a * C[0] + b * C[1] - C[2]) / (1 - a * rx - b * ry))
    d = torch.stack(depths)
    d = d * (1 + 0.0008 * torch.randn(d.shape, generator=g))                                    # ~0.5 mm noise: splits the dynamic thresholds
    out = torch.rand(d.shape, generator=g) < 0.05
    d = torch.where(out, d * (1 + 0.05 * torch.randn(d.shape, generator=g)), d)                # 5 % outliers
    d[1:, :, :4] = 0.0                                                                          # holes (filtered-out source depth)
    conf = torch.rand(d.shape, generator=g)
    return {"ref_depth": d[0][None, None].contiguous(), "srcs_depth": d[1:][None, :, None].contiguous(), "ref_conf": conf[0][None].contiguous(),
            "srcs_conf": conf[1:][None].contiguous(), "ref_cam": cams[:, 0].contiguous(), "srcs_cam": cams[:, 1:].contiguous()}


@torch.no_grad()
def f10_fusion():
    """Depth-map filtering: misc/fusion.py called the way test.py:388-409 ("pcd") and
    test.py:455-483 ("dpcd") call it.  The reference helpers hard-code ``.cuda()`` (fusion.py:9-10); with no GPU in this
    container the method is patched to a no-op for the duration of the call - the arithmetic is untouched."""
    from misc import fusion as R
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        x = fusion_inputs()
        conf_t, thres_disp, thres_view = 0.3, 1.0, 2
        # ---- static, test.py:388-409 ----
        sd = x["srcs_depth"].clone()
        for i in range(sd.shape[1]):
            sd[:, i] *= (x["srcs_conf"][:, i] > conf_t).float().unsqueeze(1)
        prob_mask = x["ref_conf"] > conf_t
        xyd, in_range = R.get_reproj(x["ref_depth"], sd, x["ref_cam"], x["srcs_cam"])
        vis_masks, vis_mask = R.vis_filter(x["ref_depth"], xyd, in_range, thres_disp, 0.01, thres_view)
        ave = R.ave_fusion(x["ref_depth"], xyd, vis_masks)
        mask = R.bin_op_reduce([prob_mask.reshape(vis_mask.shape), vis_mask], torch.min)
        idx_img = R.get_pixel_grids(*ave.size()[-2:]).unsqueeze(0)
        points = R.idx_cam2world(R.idx_img2cam(idx_img, ave, x["ref_cam"]), x["ref_cam"])[..., :3, 0].permute(0, 3, 1, 2)
        out = dict(x, conf_thresh=np.float32(conf_t), thres_disp=np.float32(thres_disp), thres_view=np.int32(thres_view),
                   s_reproj_xyd=xyd, s_in_range=in_range, s_vis_masks=vis_masks, s_geo_mask=vis_mask, s_depth=ave, s_mask=mask, s_points=points)
        # ---- dynamic, test.py:455-483 ----
        v = x["srcs_depth"].shape[1]
        dy_range = v + 1
        xyd = R.get_reproj_dynamic(x["ref_depth"], x["srcs_depth"], x["ref_cam"], x["srcs_cam"])
        vis_masks, vis_mask = R.vis_filter_dynamic(x["ref_depth"], xyd, dist_base=4, rel_diff_base=1300)
        reproj_depth = xyd[:, :, -1].clone()
        reproj_depth[~vis_mask.squeeze(2)] = 0
        geo_mask_sums = vis_masks.sum(dim=1)
        geo_mask_sum = vis_mask.sum(dim=1)
        ave = (torch.sum(reproj_depth, dim=1, keepdim=True) + x["ref_depth"]) / (geo_mask_sum + 1)
        geo_mask = geo_mask_sum >= dy_range
        for i in range(2, dy_range):
            geo_mask = torch.logical_or(geo_mask, geo_mask_sums[:, i - 2] >= i)
        mask = R.bin_op_reduce([prob_mask.reshape(geo_mask.shape), geo_mask], torch.min)
        idx_img = R.get_pixel_grids(*ave.size()[-2:]).unsqueeze(0)
        points = R.idx_cam2world(R.idx_img2cam(idx_img, ave, x["ref_cam"]), x["ref_cam"])[..., :3, 0].permute(0, 3, 1, 2)
        out.update(d_reproj_xyd=xyd, d_vis_masks=vis_masks, d_geo_mask=geo_mask, d_depth=ave, d_mask=mask, d_points=points)
        print("   static: in_range %.2f geo %.2f final %.2f | dynamic: last-threshold %.2f geo %.2f final %.2f" % (
            float(in_range.mean()), float(out["s_geo_mask"].float().mean()), float(out["s_mask"].float().mean()),
            float(vis_mask.float().mean()), float(geo_mask.float().mean()), float(mask.float().mean())))
        npz("f10_fusion.npz", **out)
    finally:
        torch.Tensor.cuda = orig


def f11_formats():
    """Bytes of a PFM depth map and a _cam.txt written by the reference's own writers (datasets/data_io.py:40-67, and
    write_cam test.py:149-166 / read_camera_parameters test.py:102-112 - test.py itself is not importable here (cv2), so
    those two small functions are executed from its source text), plus what its readers return for them."""
    import ast
    import tempfile
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_data_io", os.path.join(REF, "datasets", "data_io.py"))   # the package __init__ needs torchvision
    ref_io = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref_io)
    read_pfm, save_pfm = ref_io.read_pfm, ref_io.save_pfm
    src = open(os.path.join(REF, "test.py")).read()
    ns = {"np": np}
    for node in ast.parse(src).body:
        if isinstance(node, ast.FunctionDef) and node.name in ("write_cam", "read_camera_parameters"):
            exec(compile(ast.Module([node], []), "test.py", "exec"), ns)
    rs = np.random.RandomState(11)
    depth = (rs.rand(6, 9).astype(np.float32) * 500 + 425)
    cam = np.zeros((2, 4, 4), np.float32)
    cam[0] = np.eye(4, dtype=np.float32) + rs.randn(4, 4).astype(np.float32) * 0.1
    cam[1, :3, :3] = [[361.54125, 0, 82.900625], [0, 360.3975, 66.383875], [0, 0, 1]]
    cam[1, 3] = [425.0, 2.65, 192.0, 931.15]
    with tempfile.TemporaryDirectory() as t:
        save_pfm(os.path.join(t, "a.pfm"), depth)
        ns["write_cam"](os.path.join(t, "a_cam.txt"), cam)
        K, E = ns["read_camera_parameters"](os.path.join(t, "a_cam.txt"))
        npz("f11_formats.npz", depth=depth, cam=cam, pfm_bytes=np.frombuffer(open(os.path.join(t, "a.pfm"), "rb").read(), np.uint8),
            cam_bytes=np.frombuffer(open(os.path.join(t, "a_cam.txt"), "rb").read(), np.uint8),
            depth_read_by_reference=np.ascontiguousarray(read_pfm(os.path.join(t, "a.pfm"))[0]), K_read_by_reference=K, E_read_by_reference=E)


@torch.no_grad()
def f20_feature_heads():
    """SURVEY.md section 8f #4, producer side: the LAST 3x3 convolutions of the reference's feature side, inputs and outputs captured with
    forward hooks while the reference's own modules run - FMT_with_pathway.smooth_1/2/3 (FMT.py:195-197, called per view at FMT.py:231-233)
    inside a full FMT_with_pathway.forward on the shipped FMT_config, and FPNDecoder.out1/2/3 (module.py:247-270) inside FPNDecoder.forward."""
    from models.FMT import FMT_with_pathway
    from models.module import FPNDecoder
    cfg = json.load(open(os.path.join(REF, "config", "mvsformer++.json")))["arch"]["args"]
    g = torch.Generator().manual_seed(20)
    arrs = {}
    # ---- FMT pathway: stages 2-4 of the shipped model ----
    fmt = FMT_with_pathway(**cfg["FMT_config"]).eval()
    seed_weights(fmt, 41)
    B, V = 1, 3
    feats = {"stage1": torch.randn(B, V, 64, 3, 9, generator=g), "stage2": torch.randn(B, V, 32, 5, 18, generator=g),
             "stage3": torch.randn(B, V, 16, 10, 36, generator=g), "stage4": torch.randn(B, V, 8, 20, 72, generator=g)}     # 72 > one 64-wide tile, ragged
    cap = {1: [], 2: [], 3: []}
    hooks = [getattr(fmt, "smooth_%d" % k).register_forward_hook(lambda m, i, o, k=k: cap[k].append((i[0].clone(), o.clone()))) for k in (1, 2, 3)]
    out = fmt(feats)
    for h in hooks:
        h.remove()
    for k in (1, 2, 3):
        assert len(cap[k]) == V
        arrs["fmt%d_x" % k] = torch.stack([c[0] for c in cap[k]], 1)            # [B, V, C, H, W]: the conv's input, view by view
        arrs["fmt%d_y" % k] = torch.stack([c[1] for c in cap[k]], 1)
        arrs["fmt%d_w" % k] = getattr(fmt, "smooth_%d" % k).weight
        assert torch.equal(arrs["fmt%d_y" % k], out["stage%d" % (k + 1)]), "smooth_k's outputs ARE the stage features the cost volume reads"
    # ---- FPN decoder heads ----
    dec = FPNDecoder(cfg["feat_chs"]).eval()
    seed_weights(dec, 42)
    ins = [torch.randn(1, 8, 24, 72, generator=g), torch.randn(1, 16, 12, 36, generator=g), torch.randn(1, 32, 6, 18, generator=g),
           torch.randn(1, 64, 3, 9, generator=g)]
    capd = {}
    hooks = [getattr(dec, "out%d" % k).register_forward_hook(lambda m, i, o, k=k: capd.__setitem__(k, (i[0].clone(), o.clone()))) for k in (1, 2, 3)]
    dec(*ins)
    for h in hooks:
        h.remove()
    for k in (1, 2, 3):
        seq = getattr(dec, "out%d" % k)
        arrs["fpn%d_x" % k], arrs["fpn%d_y" % k] = capd[k]
        arrs["fpn%d_w" % k], arrs["fpn%d_b" % k] = seq[0].weight, seq[0].bias
        for name in ("weight", "bias", "running_mean", "running_var"):
            arrs["fpn%d_bn_%s" % (k, name)] = getattr(seq[1], name)
        arrs["fpn%d_bn_eps" % k] = np.float64(seq[1].eps)
    npz("f20_feature_heads.npz", **arrs)


if __name__ == "__main__":
    only = sys.argv[1:]
    if only:
        for name in only:
            globals()[name]()
        sys.exit(0)
    f10_fusion()
    f11_formats()
    f7_transformer()
    f8_stage_transformer()
    f9_cascade_shipped()
    f1_warp()
    f2_stage()
    f3_regnets()
    f4_cascade()
    f5_small_fns()
    f6_train_mode()
    f12_train_backward()
    f13_train_backward_transformer()
    f14_stage_bd_hypotheses()
    f15_stage_other_groups()
    f16_regnet_inner()
    f17_range_variants()
    f18_costregnet2d()
    f19_position_encoding()
    f20_feature_heads()
    pin_weights()
