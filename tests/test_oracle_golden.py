"""CPU: pin the oracle (oracle/ref_path.py) against golden vectors generated from the imported reference
(tests/golden/make_golden.py).  This is what licenses using the oracle as the parity checker on the GPU box."""
import pytest
import torch

from conftest import golden_weights, load_golden, rel_l1
from oracle import ref_path as O

TOL = 2e-6   # fp32 op-for-op restatement; observed <= 1e-6 (values O(1))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_f1_warp(tag):
    fx = load_golden("f1_warp_%s.npz" % tag)
    for dv, wk, mk in (("dv2", "warped2", "mask2"), ("dv4", "warped4", "mask4")):
        w, m = O.homo_warping_3D_with_mask(fx["src_fea"], fx["src_proj"], fx["ref_proj"], fx[dv])
        assert torch.equal(m, fx[mk])
        assert (w - fx[wk]).abs().max() <= TOL
    # the fixture must really exercise out-of-frame / behind-camera voxels
    assert fx["mask4"].float().mean() > 0.02


@pytest.mark.parametrize("tag", ["s1", "s3"])
def test_f2_stage(tag):
    fx = load_golden("f2_stage_%s.npz" % tag)
    sd = golden_weights(fx)
    out = O.stage_forward(fx["features"], fx["proj"], fx["hyp"], float(fx["tmp"]), sd, G=8, return_intermediates=True)
    assert (out["volume_mean"] - fx["volume_mean"]).abs().max() <= 1e-5
    assert (out["prob_volume_pre"] - fx["prob_volume_pre"]).abs().max() <= 1e-4
    assert (out["prob_volume"] - fx["prob_volume"]).abs().max() <= 1e-5
    assert (out["photometric_confidence"] - fx["photometric_confidence"]).abs().max() <= 1e-5
    assert rel_l1(out["depth"], fx["depth"]) <= 1e-6


def test_f14_stage_bd_hypotheses():
    """[B, D] hypotheses: the oracle broadcasts them exactly as the reference's warp / depth_regression do (warping.py:91, module.py:650-652)."""
    fx = load_golden("f14_stage_bd_hyp.npz")
    sd = golden_weights(fx)
    assert fx["hyp"].dim() == 2
    out = O.stage_forward(fx["features"], fx["proj"], fx["hyp"], 1.0, sd, G=8)
    assert (out["prob_volume_pre"] - fx["prob_volume_pre"]).abs().max() <= 1e-4
    assert (out["prob_volume"] - fx["prob_volume"]).abs().max() <= 1e-5
    assert rel_l1(out["depth"], fx["depth"]) <= 1e-6


@pytest.mark.parametrize("name,fn", [("f3_costregnet.npz", O.cost_regnet), ("f3_costregnet3d_d4.npz", O.cost_regnet3d),
                                     ("f3_costregnet3d_d8.npz", O.cost_regnet3d)])
def test_f3_regnets(name, fn):
    fx = load_golden(name)
    sd = {"cost_reg." + k: v for k, v in golden_weights(fx).items()}
    y = fn(fx["x"], sd)
    assert y.shape == fx["y"].shape
    assert (y - fx["y"]).abs().max() <= 1e-4 * max(1.0, float(fx["y"].abs().max()))


@pytest.mark.parametrize("tag", ["g4_s1", "g4_s3", "g16_s2"])
def test_f15_stage_other_groups(tag):
    """base_ch != 8 (cost_volume.py:29-49): G groups in the correlation, CostRegNet(G, G) / CostRegNet3D(G, G) widths."""
    fx = load_golden("f15_stage_%s.npz" % tag)
    sd = golden_weights(fx)
    out = O.stage_forward(fx["features"], fx["proj"], fx["hyp"], float(fx["tmp"]), sd, G=int(fx["base_ch"]), return_intermediates=True)
    assert out["volume_mean"].shape[1] == int(fx["base_ch"])
    assert (out["volume_mean"] - fx["volume_mean"]).abs().max() <= 1e-5
    assert (out["prob_volume_pre"] - fx["prob_volume_pre"]).abs().max() <= 1e-4
    assert (out["prob_volume"] - fx["prob_volume"]).abs().max() <= 1e-5
    assert rel_l1(out["depth"], fx["depth"]) <= 1e-6


@pytest.mark.parametrize("tag", ["crn_4_8", "crn_12_4", "crn_8_8_nolast", "crn3d_12_8", "crn3d_6_6", "crn3d_8_8_logvar"])
def test_f16_regnet_inner(tag):
    """in_channels != base_channels (`inner`, module.py:385-388 / 481-484), last_layer=False, log_var=True."""
    fx = load_golden("f16_regnet_inner.npz")
    sd = {"cost_reg." + k: v for k, v in golden_weights(fx, prefix=tag + ".w.").items()}
    y = (O.cost_regnet3d if "3d" in tag else O.cost_regnet)(fx[tag + ".x"], sd)
    assert y.shape == fx[tag + ".y"].shape
    assert (y - fx[tag + ".y"]).abs().max() <= 1e-4 * max(1.0, float(fx[tag + ".y"].abs().max()))


@pytest.mark.parametrize("tag", ["b8", "b4"])
def test_f18_costregnet2d(tag):
    fx = load_golden("f18_costregnet2d.npz")
    sd = {"cost_reg." + k: v for k, v in golden_weights(fx, prefix=tag + ".w.").items()}
    y = O.cost_regnet2d(fx[tag + ".x"], sd)
    assert y.shape == fx[tag + ".y"].shape
    assert (y - fx[tag + ".y"]).abs().max() <= 1e-4 * max(1.0, float(fx[tag + ".y"].abs().max()))


def test_f19_position_encoding():
    fx = load_golden("f19_position_encoding.npz")
    B, D, H, W = fx["hyp"].shape
    pos, hmin, hmax, wmin, wmax = O.get_position_3d(H, W, fx["K"], fx["hyp"], 425.0, 935.0)
    assert torch.allclose(pos, fx["position3d"], rtol=1e-5, atol=1e-6)
    assert torch.allclose(torch.stack([hmin, hmax, wmin, wmax]), fx["ranges"], rtol=1e-6)
    for C, rescale in ((8, 4.0), (6, 2.5)):
        assert (O.position_encoding_3d(fx["position3d"], C, rescale) - fx["pe_c%d" % C]).abs().max() <= 1e-6


def test_f20_feature_heads():
    """the oracle's restatement of the feature side's last 3x3 convolutions against what the reference's own
    FMT_with_pathway / FPNDecoder modules produced (forward hooks, tests/golden/make_golden.py:f20_feature_heads)."""
    fx = load_golden("f20_feature_heads.npz")
    for k in (1, 2, 3):
        x, y = fx["fmt%d_x" % k], fx["fmt%d_y" % k]
        B, V = x.shape[:2]
        got = O.feature_head(x.flatten(0, 1), fx["fmt%d_w" % k]).view_as(y)
        assert (got - y).abs().max() <= 1e-5 * max(1.0, float(y.abs().max()))
        bn = {n: fx["fpn%d_bn_%s" % (k, n)] for n in ("weight", "bias", "running_mean", "running_var")}
        bn["eps"] = fx["fpn%d_bn_eps" % k]
        got = O.feature_head(fx["fpn%d_x" % k], fx["fpn%d_w" % k], fx["fpn%d_b" % k], bn, swish=True)
        assert (got - fx["fpn%d_y" % k]).abs().max() <= 1e-5 * max(1.0, float(fx["fpn%d_y" % k].abs().max()))


def test_f4_cascade():
    fx = load_golden("f4_cascade.npz")
    feats = {"stage%d" % s: fx["features%d" % s] for s in range(1, 5)}
    projs = {"stage%d" % s: fx["proj%d" % s] for s in range(1, 5)}
    sds = [golden_weights(fx, "w%d." % s) for s in range(1, 5)]
    out = O.cascade_forward(feats, projs, fx["depth_values"], sds, ndepths=[32, 16, 8, 4],
                            depth_interals_ratio=[4.0, 2.67, 1.5, 1.0], base_ch=[8, 8, 8, 8])
    for s in range(1, 5):
        st = out["stage%d" % s]
        assert rel_l1(st["depth_values"], fx["hyp%d" % s]) <= 1e-6
        assert rel_l1(st["depth"], fx["depth%d" % s]) <= 1e-6
        assert (st["photometric_confidence"] - fx["conf%d" % s]).abs().max() <= 1e-4
    assert rel_l1(out["refined_depth"], fx["refined_depth"]) <= 1e-6
    assert (out["photometric_confidence"] - fx["photometric_confidence"]).abs().max() <= 1e-4


def test_f17_range_variants():
    fx = load_golden("f17_range_variants.npz")
    px = fx["pixel_ranges"]
    assert torch.equal(O.init_range(px, 8, 5, 6), fx["init_range_pixel"])
    assert torch.equal(O.init_inverse_range(px, 8, 5, 6), fx["init_inverse_range_pixel"])
    got = O.schedule_inverse_range(fx["prev_depth"], fx["prev_hyp"], 4, 1.0, 10, 12, shift=True)
    assert torch.allclose(got, fx["schedule_inverse_range_shift"], rtol=1e-6, atol=0)
    got = O.schedule_range(fx["prev_depth"], 4, fx["schedule_range_itv_pixel"], 10, 12)
    assert torch.allclose(got, fx["schedule_range_pixel"], rtol=1e-6, atol=0)


def test_f5_small_fns():
    fx = load_golden("f5_small_fns.npz")
    for D, n in ((32, 4), (16, 3), (8, 2)):
        assert torch.allclose(O.depth_regression(fx["p%d" % D], fx["dv%d" % D]), fx["dreg%d" % D], rtol=1e-6, atol=0)
        assert torch.allclose(O.conf_regression(fx["p%d" % D], n=n), fx["conf%d_n%d" % (D, n)], rtol=1e-6, atol=1e-7)
    dv = fx["depth_values"]
    assert torch.equal(O.init_range(dv, 8, 5, 6), fx["init_range"])
    assert torch.equal(O.init_inverse_range(dv, 8, 5, 6), fx["init_inverse_range"])
    got = O.schedule_inverse_range(fx["prev_depth"], fx["prev_hyp"], 4, 2.67, 10, 12)
    assert torch.allclose(got, fx["schedule_inverse_range"], rtol=1e-6, atol=0)
    got = O.schedule_range(fx["prev_depth"], 4, fx["schedule_range_itv"], 10, 12)
    assert torch.allclose(got, fx["schedule_range"], rtol=1e-6, atol=0)


def test_f6_train_and_reg():
    fx = load_golden("f6_stage_train_ce.npz")
    out = O.stage_forward(fx["features"], fx["proj"], fx["hyp"], 5.0, golden_weights(fx), G=8, training=True)
    assert (out["depth"] != fx["depth"]).float().mean() <= 0.002      # argmax ties may flip on a rounding difference
    assert (out["prob_volume"] - fx["prob_volume"]).abs().max() <= 1e-5
    fx = load_golden("f6_stage_reg.npz")
    out = O.stage_forward(fx["features"], fx["proj"], fx["hyp"], 1.0, golden_weights(fx), G=8, depth_type="reg")
    assert rel_l1(out["depth"], fx["depth"]) <= 1e-6
    assert (out["photometric_confidence"] - fx["photometric_confidence"]).abs().max() <= 1e-5


# ---------------------------------------------------------------- stage-1 transformer regulariser
def _tcfg(fx):
    import json
    return json.loads(fx["cfg"])


def test_f7_transformer():
    fx = load_golden("f7_transformer.npz")
    cfg = _tcfg(fx)
    sd = {"cost_reg." + k: v for k, v in golden_weights(fx).items()}
    dv = fx["depth_values"]
    pos, *rng = O.get_position_3d(16, 24, fx["K"], fx["hyp"], dv.min(), dv.max())
    assert (pos - fx["position3d"]).abs().max() <= TOL
    assert (torch.stack(rng) - fx["pe_range"]).abs().max() <= 1e-4
    kw = dict(num_heads=cfg["num_heads"], train_avg_length=cfg["train_avg_length"], softmax_scale=cfg["softmax_scale"])
    y = O.pure_transformer_cost_reg(fx["x"], fx["position3d"], sd, **kw)
    assert (y - fx["y"]).abs().max() <= 2e-5 * max(1.0, float(fx["y"].abs().max()))
    y0 = O.pure_transformer_cost_reg(fx["x"], None, sd, **kw)
    assert (y0 - fx["y_nope"]).abs().max() <= 2e-5 * max(1.0, float(fx["y_nope"].abs().max()))
    assert (fx["y"] - fx["y_nope"]).abs().max() > 1e-3            # the position encoding must matter in the fixture


def test_f8_stage_transformer():
    fx = load_golden("f8_stage_transformer.npz")
    sd = golden_weights(fx)
    out = O.stage_forward(fx["features"].float(), fx["proj"], fx["hyp"], 5.0, sd, G=8, position3d=fx["position3d"],
                          transformer_config=_tcfg(fx))
    assert (out["prob_volume_pre"] - fx["prob_volume_pre"]).abs().max() <= 1e-4
    assert (out["prob_volume"] - fx["prob_volume"]).abs().max() <= 1e-5
    assert rel_l1(out["depth"], fx["depth"]) <= 1e-6


def test_f9_cascade_shipped():
    fx, f4 = load_golden("f9_cascade_shipped.npz"), load_golden("f4_cascade.npz")
    feats = {"stage%d" % s: f4["features%d" % s] for s in range(1, 5)}
    projs = {"stage%d" % s: f4["proj%d" % s] for s in range(1, 5)}
    sds = [golden_weights(fx, "w%d." % s) for s in range(1, 5)]
    out = O.cascade_forward(feats, projs, f4["depth_values"], sds, ndepths=[32, 16, 8, 4], depth_interals_ratio=[4.0, 2.67, 1.5, 1.0],
                            base_ch=[8, 8, 8, 8], use_pe3d=True, transformer_config=[_tcfg(fx)])
    for s in range(1, 5):
        assert rel_l1(out["stage%d" % s]["depth"], fx["depth%d" % s]) <= 2e-6
        assert (out["stage%d" % s]["photometric_confidence"] - fx["conf%d" % s]).abs().max() <= 1e-4
    assert rel_l1(out["refined_depth"], fx["refined_depth"]) <= 2e-6


# ---------------------------------------------------------------- depth-map filtering
def test_f10_fusion():
    from oracle import fusion_ref as FR
    fx = load_golden("f10_fusion.npz")
    s = FR.filter_depth(fx["ref_depth"], fx["ref_conf"], fx["srcs_depth"], fx["srcs_conf"], fx["ref_cam"], fx["srcs_cam"],
                        conf_thresh=fx["conf_thresh"], thres_disp=fx["thres_disp"], thres_view=fx["thres_view"])
    assert (s["reproj_xyd"] - fx["s_reproj_xyd"]).abs().max() <= 1e-4
    assert torch.equal(s["in_range"], fx["s_in_range"]) and torch.equal(s["vis_masks"], fx["s_vis_masks"])
    assert torch.equal(s["geo_mask"], fx["s_geo_mask"]) and torch.equal(s["mask"], fx["s_mask"])
    assert (s["depth"] - fx["s_depth"]).abs().max() <= 1e-4 and (s["points"] - fx["s_points"]).abs().max() <= 1e-3
    d = FR.dynamic_filter_depth(fx["ref_depth"], fx["ref_conf"], fx["srcs_depth"], fx["ref_cam"], fx["srcs_cam"], conf_thresh=fx["conf_thresh"])
    assert (d["reproj_xyd"] - fx["d_reproj_xyd"]).abs().max() <= 1e-4
    assert torch.equal(d["vis_masks"], fx["d_vis_masks"]) and torch.equal(d["geo_mask"], fx["d_geo_mask"]) and torch.equal(d["mask"], fx["d_mask"])
    assert (d["depth"] - fx["d_depth"]).abs().max() <= 1e-4 and (d["points"] - fx["d_points"]).abs().max() <= 1e-3
    assert 0.2 < float(fx["s_mask"].float().mean()) < 0.8 and 0.2 < float(fx["d_mask"].float().mean()) < 0.9     # both outcomes present
