"""CPU: the real kernel sources (csrc/*.hip) executed through the host emulator (tests/hipemu) against the oracle and
the golden vectors.  This checks indexing, tiling, MFMA fragment layouts and barrier placement without a GPU; the
numbers that count are produced by tests/test_gpu_parity.py on the MI355X."""
import pytest

import parity_cases as P
from conftest import PRECS, PRECS_ALL          # [None = the product default (the "stagemix" policy), "bf16x3" = the fp32-equivalent mode]


@pytest.mark.parametrize("tag", ["a", "b"])
def test_warp_golden(emu, tag):
    P.case_warp_golden(emu, tag)


def test_warp_dtypes(emu):
    P.case_warp_dtypes(emu)


def test_warp_siblings(emu):
    P.case_warp_siblings(emu)


def test_precisions(emu):
    P.case_precisions(emu)


@pytest.mark.parametrize("prec", PRECS)
def test_single_layers(emu, prec):
    P.case_single_layers(emu, prec)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("name", ["f3_costregnet.npz", "f3_costregnet3d_d4.npz", "f3_costregnet3d_d8.npz"])
def test_regnet_golden(emu, name, prec):
    P.case_regnet_golden(emu, name, prec)


def test_generic_conv_layers(emu):
    P.case_generic_conv_layers(emu)


def test_generic_conv_fuzz(emu):
    P.case_generic_conv_fuzz(emu, n_cases=60, seed=0)


def test_regnet_generic_golden(emu):
    P.case_regnet_generic_golden(emu)


def test_costregnet2d_golden(emu):
    P.case_costregnet2d_golden(emu)


def test_position_encoding_golden(emu):
    P.case_position_encoding_golden(emu)


@pytest.mark.parametrize("prec", PRECS)
@pytest.mark.parametrize("tag", ["g4_s1", "g4_s3", "g16_s2"])
def test_stage_other_groups_golden(emu, tag, prec):
    P.case_stage_other_groups_golden(emu, tag, prec)


def test_cascade_other_groups_vs_oracle(emu):
    P.case_cascade_other_groups_vs_oracle(emu, G=4, prec="bf16x3")


def test_stage_pieces(emu):
    P.case_stage_pieces(emu)


@pytest.mark.parametrize("prec", PRECS_ALL)
@pytest.mark.parametrize("tag", ["s1", "s3"])
def test_stage_golden(emu, tag, prec):
    P.case_stage_golden(emu, tag, prec)


@pytest.mark.parametrize("prec", PRECS)
def test_stage_bd_hypotheses(emu, prec):
    P.case_stage_bd_hypotheses(emu, prec)


@pytest.mark.parametrize("prec", PRECS)
def test_stage_modes(emu, prec):
    P.case_stage_modes(emu, prec)


@pytest.mark.parametrize("prec", PRECS)
def test_stage_lowp_features(emu, prec):
    P.case_stage_lowp_features(emu, prec)


def test_small_fns(emu):
    P.case_small_fns(emu)


def test_auto_policy(emu):
    P.case_auto_policy(emu, quick=True)


def test_feature_heads(emu):
    P.case_feature_heads(emu)


def test_fused_small_launches(emu):
    P.case_fused_small_launches(emu)


def test_range_variants(emu):
    P.case_range_variants(emu)


def test_generic_shapes(emu):
    P.case_generic_shapes(emu)


def test_vis_cnn(emu):
    P.case_vis_cnn(emu)


def test_gather_variants(emu):
    P.case_gather_variants(emu)


def test_gather_windows(emu):
    P.case_gather_windows(emu)


def test_split_format(emu):
    P.case_split_format(emu)


def test_f16_layers(emu):
    P.case_f16_layers(emu)




def test_f16_saturation(emu):
    P.case_f16_saturation(emu)


def test_slab_exchange_kernels(emu):
    P.case_slab_exchange_kernels(emu)


@pytest.mark.parametrize("prec", PRECS)        # the emulator runs the product default and the fp32-equivalent format; "f16x2" / "f16" and
def test_cascade_golden(emu, prec):            # case_f16_cascade run on the MI355X (test_gpu_parity.py) - the CPU suite's time budget
    P.case_cascade_golden(emu, prec)


@pytest.mark.parametrize("attn", [None, "bf16x3"])
def test_transformer_golden(emu, attn):
    P.case_transformer_golden(emu, attn)


@pytest.mark.parametrize("attn", [None, "bf16x3"])
def test_stage_transformer_golden(emu, attn):
    P.case_stage_transformer_golden(emu, attn)


@pytest.mark.parametrize("prec,attn", [(None, None), ("bf16x3", "bf16x3")])
def test_cascade_shipped_golden(emu, prec, attn):
    P.case_cascade_shipped_golden(emu, conv_precision=prec, attention_precision=attn)


def test_attention_stress(emu):
    P.case_attention_stress(emu)


@pytest.mark.parametrize("variant", [0, 1, 3, 4, 5, 80])
def test_attention_stress_f16(emu, variant, monkeypatch):
    """The 16-bit flash attention (module default) incl. its tile-shape variants: masked tail, rescales far into the key stream."""
    monkeypatch.setenv("MVS_ATTN_VARIANT", str(variant))
    P.case_attention_stress(emu, n=300 if variant else 200, mode="attn16")


def test_attention_overflow(emu):
    P.case_attention_overflow(emu)


def test_fusion_golden(emu):
    P.case_fusion_golden(emu)


def test_attention_bf16p(emu):
    P.case_attention_stress(emu, bf16p=True)
    P.case_stage_transformer_bf16p(emu)


def test_aggregate_backward(emu):
    P.case_aggregate_backward(emu)


@pytest.mark.parametrize("tag", ["s3", "s1"])
def test_train_backward_golden(emu, tag):
    P.case_train_backward_golden(emu, tag)


def test_train_path_properties(emu):
    P.case_train_path_properties(emu)


def test_train_kernels(emu):
    P.case_train_kernels(emu)


def test_attention_backward(emu):
    P.case_attention_backward(emu)


def test_transformer_block_backward(emu):
    P.case_transformer_block_backward(emu)


def test_regnet_train_recompute(emu):
    P.case_regnet_train_recompute(emu)


def test_regnet_train_native(emu):
    P.case_regnet_train_native(emu)


def test_train_backward_transformer_golden(emu):
    P.case_train_backward_transformer_golden(emu)


def test_device_packing(emu):
    P.case_device_packing(emu)


def test_train_fp32_configured_head(emu):
    P.case_train_fp32_configured_head(emu)
