"""Mirror of the reference's ``models/warping.py`` warping entry points: ``homo_warping_3D_with_mask`` (the one on the hot path,
warping.py:69-109) and its two siblings ``homo_warping_3D`` (:152-189, no mask) and ``diff_homo_warping_3D_with_mask`` (:112-149, the
same forward values; it differs only in letting autograd see the sampling grid).  All three are forward forms on one HIP kernel."""
from __future__ import annotations

import torch

from . import ops
from .module import _no_grad_path


def homo_warping_3D_with_mask(src_fea: torch.Tensor, src_proj: torch.Tensor, ref_proj: torch.Tensor, depth_values: torch.Tensor):
    """Warp ``src_fea [B,C,H,W]`` onto the D fronto-parallel planes of the reference frustum.

    Same arguments and results as the reference (models/warping.py:69-109): ``src_proj``/``ref_proj`` are the
    composed 4x4 projections, ``depth_values`` is ``[B,D]`` or ``[B,D,H,W]``; returns
    ``(warped [B,C,D,H,W] fp32, proj_mask [B,D,H,W] bool)`` where the mask marks hypotheses that project outside
    the source image or behind the source camera.  One HIP kernel: bilinear, zeros padding, align_corners=True.

    The fused ``StageNet`` never calls this (it does not materialise the warped volume); it exists for callers
    and tests that want the reference's intermediate.
    """
    _no_grad_path(src_fea)
    hom = ops.homography_from_proj(src_proj, ref_proj)
    return ops.homo_warp(src_fea, hom, depth_values, True, True)


def homo_warping_3D(src_fea: torch.Tensor, src_proj: torch.Tensor, ref_proj: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    """``warped [B,C,D,H,W]`` only (reference models/warping.py:152-189): the same kernel with the mask output switched off."""
    _no_grad_path(src_fea)
    hom = ops.homography_from_proj(src_proj, ref_proj)
    return ops.homo_warp(src_fea, hom, depth_values, True, False)[0]


def diff_homo_warping_3D_with_mask(src_fea: torch.Tensor, src_proj: torch.Tensor, ref_proj: torch.Tensor, depth_values: torch.Tensor):
    """Forward of the reference's grid-differentiable variant (models/warping.py:112-149): identical values to
    ``homo_warping_3D_with_mask``; as an inference form it refuses tensors that require grad (no caller in the reference tree
    differentiates through it: ``StageNet`` uses the no-grad-grid form, cost_volume.py:72)."""
    _no_grad_path(src_fea, src_proj, ref_proj, depth_values)
    hom = ops.homography_from_proj(src_proj, ref_proj)
    return ops.homo_warp(src_fea, hom, depth_values, True, True)
