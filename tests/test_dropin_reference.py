"""Build-container only (needs the read-only reference checkout): the REFERENCE's own top-level model, shipped config, with its
``fusions`` swapped by ``patch_model`` - the one-line integration of INTEGRATION.md - must reproduce the unpatched model.  The HIP
kernels run through the host emulator; backbone, FMT, the cascade loop and get_position_3d stay the reference's code.  Skipped
wherever /root/reference does not exist (e.g. on the GPU box)."""
import copy
import json
import os
import sys

import pytest
import torch

REF = os.environ.get("MVS_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "models")), reason="reference checkout not present")


@pytest.mark.parametrize("prec", [None, "bf16x3"])
def test_patch_model_on_the_reference_network(emu, prec):
    """prec None = the product defaults (policy "stagemix": fp32-equivalent coarse stages, fp16 regulariser activations on the fine ones; 16-bit attention operands): 5e-4 on depth; "bf16x3" = the
    fp32-equivalent forms of both: 2e-4 (the residue is the feature extractor's own fp32 summation order)."""
    sys.path.insert(0, REF)
    try:
        from models.networks.DINOv2_mvsformer_model import DINOv2MVSNet
    finally:
        sys.path.remove(REF)
    from mvsformerplusplus_amd import patch_model, synth
    from mvsformerplusplus_amd.cost_volume import StageNet as HipStageNet
    args = json.load(open(os.path.join(REF, "config", "mvsformer++.json")))["arch"]["args"]
    torch.manual_seed(0)
    model = DINOv2MVSNet(args).eval()
    synth.randomize_bn_(model, seed=3)
    H, W, V = 64, 128, 3
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(1, V, 3, H, W, generator=g)
    cams = synth.make_cameras(V, H, W, baseline=30.0, rot_deg=1.0, seed=2)
    projs = synth.stage_proj_matrices(cams, 4)
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None]
    with torch.no_grad():
        ref = model(imgs, projs, dv)
        patched = patch_model(copy.deepcopy(model), conv_precision=prec, attention_precision=prec)
        assert all(isinstance(f, HipStageNet) for f in patched.fusions)
        assert patched.fusions[0].cost_reg.kind == "transformer"           # the shipped stage-1 regulariser
        out = patched(imgs, projs, dv)
    assert set(out.keys()) == set(ref.keys())
    for s in range(1, 5):
        a, b = out["stage%d" % s]["depth"], ref["stage%d" % s]["depth"]
        assert float(((a - b).abs() / b.abs()).mean()) <= (2e-4 if prec else 5e-4), s
    r = float(((out["refined_depth"] - ref["refined_depth"]).abs() / ref["refined_depth"].abs()).mean())
    assert r <= (2e-4 if prec else 5e-4), r
    assert (out["photometric_confidence"] - ref["photometric_confidence"]).abs().max() <= (5e-3 if prec else 3e-2)


def test_patch_model_on_the_reference_network_other_base_ch(emu):
    """The reference's own network built with base_ch = 4 (not a shipped value) and all-"Normal" regularisers: cost_volume.py:29-49 then
    builds CostRegNet(4, 4) / CostRegNet3D(4, 4).  patch_model must take the model as it is (strict state-dict load) and reproduce it -
    the stages run the direct gather with 4 groups and the shape-generic exact-fp32 convolution kernel (ABI v10)."""
    sys.path.insert(0, REF)
    try:
        from models.networks.DINOv2_mvsformer_model import DINOv2MVSNet
    finally:
        sys.path.remove(REF)
    from mvsformerplusplus_amd import patch_model, synth
    args = json.load(open(os.path.join(REF, "config", "mvsformer++.json")))["arch"]["args"]
    args["base_ch"] = [4, 4, 4, 4]
    args["cost_reg_type"] = ["Normal"] * 4
    torch.manual_seed(0)
    model = DINOv2MVSNet(args).eval()
    synth.randomize_bn_(model, seed=4)
    assert model.fusions[0].cost_reg.conv1.conv.weight.shape[:2] == (8, 4)
    H, W, V = 64, 128, 3
    imgs = torch.rand(1, V, 3, H, W, generator=torch.Generator().manual_seed(1))
    cams = synth.make_cameras(V, H, W, baseline=30.0, rot_deg=1.0, seed=2)
    projs = synth.stage_proj_matrices(cams, 4)
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None]
    with torch.no_grad():
        ref = model(imgs, projs, dv)
        patched = patch_model(copy.deepcopy(model), conv_precision="bf16x3")
        assert all(f._generic_regulariser() for f in patched.fusions)
        out = patched(imgs, projs, dv)
    for s in range(1, 5):
        a, b = out["stage%d" % s]["depth"], ref["stage%d" % s]["depth"]
        assert float(((a - b).abs() / b.abs()).mean()) <= 2e-4, s
    r = float(((out["refined_depth"] - ref["refined_depth"]).abs() / ref["refined_depth"].abs()).mean())
    assert r <= 2e-4, r
    assert (out["photometric_confidence"] - ref["photometric_confidence"]).abs().max() <= 5e-3


def test_patch_model_trains_like_the_reference_network(emu):
    """train.py's use: the reference's whole network in .train() mode with its stages swapped by patch_model.  The stage-1 loss
    (before any argmax-dependent hypothesis scheduling) must give the same gradients for the stage's own parameters AND for the
    reference's feature extractor upstream of it; the later stages must produce finite gradients for everything."""
    sys.path.insert(0, REF)
    try:
        from models.networks.DINOv2_mvsformer_model import DINOv2MVSNet
    finally:
        sys.path.remove(REF)
    from mvsformerplusplus_amd import patch_model, synth
    args = json.load(open(os.path.join(REF, "config", "mvsformer++.json")))["arch"]["args"]
    torch.manual_seed(0)
    model = DINOv2MVSNet(args).train()
    synth.randomize_bn_(model, seed=3)
    patched = patch_model(copy.deepcopy(model))
    assert all(f.training for f in patched.fusions)
    H, W, V = 64, 128, 3
    g = torch.Generator().manual_seed(1)
    imgs = torch.rand(1, V, 3, H, W, generator=g)
    cams = synth.make_cameras(V, H, W, baseline=30.0, rot_deg=1.0, seed=2)
    projs = synth.stage_proj_matrices(cams, 4)
    dv = torch.arange(425.0, 2.65 * 191.5 + 425.0, 2.65)[None]
    grads = []
    for m in (model, patched):
        torch.manual_seed(5)                                          # any stochastic layer of the backbone draws the same numbers
        out = m(imgs, projs, dv)
        pre = out["stage1"]["prob_volume_pre"]
        R = torch.randn(pre.shape, generator=torch.Generator().manual_seed(9))
        loss = (torch.softmax(pre, 1) * R).sum() + sum(out["stage%d" % s]["prob_volume_pre"].square().mean() for s in (2, 3, 4)) * 0.0
        loss.backward()
        grads.append({n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None})
    ref_g, hip_g = grads
    assert set(ref_g) == set(hip_g)
    # parameters whose gradient is analytically zero (a bias in front of a normalisation, a constant added to every logit under
    # the softmax) hold rounding noise in both models: they are compared on an absolute floor, not relative to themselves
    top = max(float(gr.abs().max()) for gr in ref_g.values())
    worst, n_cmp = 0.0, 0
    for n, gr in ref_g.items():
        if float(gr.abs().max()) <= 1e-6 * top:
            assert float(hip_g[n].abs().max()) <= 1e-5 * top, n
            continue
        worst = max(worst, float((hip_g[n] - gr).abs().max() / gr.abs().max()))
        n_cmp += 1
    if os.environ.get("MVS_TEST_VERBOSE"):
        errs = sorted(((float((hip_g[n] - gr).abs().max() / gr.abs().max()), n) for n, gr in ref_g.items() if float(gr.abs().max()) > 1e-6 * top), reverse=True)
        for e, n in errs[:12] + errs[-3:]:
            print("%.3e  %s" % (e, n))
    assert n_cmp > 250 and worst <= 2e-3, (n_cmp, worst)             # measured: 7e-5 worst over 284 tensors
    assert any(n.startswith("fusions.0.cost_reg.attention_layers") for n in ref_g) and any(not n.startswith("fusions.") for n in ref_g)
    # all four stages in the loss (stages 2-4: the native U-Net training path): every parameter that receives a gradient gets a finite one
    patched.zero_grad()
    out = patched(imgs, projs, dv)
    sum(out["stage%d" % s]["prob_volume_pre"].square().mean() for s in (1, 2, 3, 4)).backward()
    got = {n: p.grad for n, p in patched.named_parameters() if p.grad is not None}
    assert all(torch.isfinite(g).all() for g in got.values())
    for s in (1, 2, 3):
        assert float(got["fusions.%d.cost_reg.conv1.conv.weight" % s].abs().max()) > 0
