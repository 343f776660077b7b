"""CPU: the plain-C operator restatement (oracle/stage_ref.c + oracle/c_path.py) against the golden vectors produced by
the reference and against the torch restatement - two independent oracles must agree before either is trusted."""
import numpy as np
import pytest
import torch

from conftest import golden_weights, load_golden, rel_l1
from oracle import c_path


@pytest.mark.parametrize("tag", ["a", "b"])
def test_c_warp_vs_reference_golden(tag):
    fx = load_golden("f1_warp_%s.npz" % tag)
    for b in range(fx["src_fea"].shape[0]):
        M = (fx["src_proj"][b] @ torch.inverse(fx["ref_proj"][b])).numpy()
        hom = np.concatenate([M[:3, :3].reshape(9), M[:3, 3]]).astype(np.float32)
        w, m = c_path.homo_warp(fx["src_fea"][b].numpy(), hom, fx["dv4"][b].numpy())
        assert np.abs(w - fx["warped4"][b].numpy()).max() <= 2e-4
        assert (m != fx["mask4"][b].numpy()).mean() <= 2e-3


@pytest.mark.parametrize("tag", ["s3", "s1"])
def test_c_stage_vs_reference_golden(tag):
    fx = load_golden("f2_stage_%s.npz" % tag)
    sd = golden_weights(fx)
    out = c_path.stage_forward(fx["features"][0].numpy(), fx["proj"][0].numpy(), fx["hyp"][0].numpy(), float(fx["tmp"]), sd)
    assert np.abs(out["volume_mean"] - fx["volume_mean"][0].numpy()).max() <= 5e-5
    assert np.abs(out["prob_volume_pre"] - fx["prob_volume_pre"][0].numpy()).max() <= 1e-3
    assert rel_l1(torch.from_numpy(out["depth"]), fx["depth"][0]) <= 2e-5
    assert np.abs(out["photometric_confidence"] - fx["photometric_confidence"][0].numpy()).max() <= 2e-4


def test_c_upsample_matches_torch_trilinear():
    g = torch.Generator().manual_seed(0)
    x = torch.rand(3, 5, 7, generator=g)
    y = np.empty((3, 10, 14), np.float32)
    c_path.lib().ref_upsample_bilinear_ac(c_path._p(x.numpy()), c_path._p(y), 3, 5, 7, 10, 14)
    ref = torch.nn.functional.interpolate(x[None, None], [3, 10, 14], mode="trilinear", align_corners=True)[0, 0]
    assert np.abs(y - ref.numpy()).max() <= 1e-6
