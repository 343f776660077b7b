"""Mirror of the reference's ``models/warping.py`` entry point used on the hot path."""
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
