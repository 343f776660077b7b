"""MI355X (gfx950) implementation of MVSFormer++'s depth-inference hot path: homography warp -> group-wise correlation
cost volume -> 3D-conv regularisation -> depth regression, behind the reference's ``nn.Module`` API.

    from mvsformerplusplus_amd import patch_model, CascadeDepthHead, StageNet

The HIP library (``csrc/libmvs_hip.so``, built by ``python -m mvsformerplusplus_amd.build``) is loaded on first use;
there is no CPU or eager-PyTorch fallback.  See DESIGN.md and INTEGRATION.md.
"""
from .cascade import CascadeDepthHead, patch_model
from .cost_volume import StageNet
from .module import (Conv3d, ConvBnReLU, CostRegNet, CostRegNet2D, CostRegNet3D, Deconv3d, PureTransformerCostReg, conf_regression, depth_regression,
                     init_inverse_range, init_range, schedule_inverse_range, schedule_range)
from . import fusion
from .ops import PackedFeatures, pack_features
from .handoff import TiledFeatureHead
from .position_encoding import PositionEncoding3D, get_position_3d
from .warping import diff_homo_warping_3D_with_mask, homo_warping_3D, homo_warping_3D_with_mask

__all__ = ["CascadeDepthHead", "patch_model", "StageNet", "Conv3d", "Deconv3d", "ConvBnReLU", "CostRegNet", "CostRegNet3D", "CostRegNet2D",
           "PureTransformerCostReg", "get_position_3d", "PositionEncoding3D", "fusion", "PackedFeatures", "pack_features", "TiledFeatureHead",
           "depth_regression", "conf_regression", "init_range", "init_inverse_range", "schedule_inverse_range", "schedule_range",
           "homo_warping_3D_with_mask", "homo_warping_3D", "diff_homo_warping_3D_with_mask"]
