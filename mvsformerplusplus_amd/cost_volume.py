"""Mirror of the reference's ``models/cost_volume.py``: one cascade stage behind the same ``nn.Module`` API.

``StageNet(args, ndepth, stage_idx).forward(features, proj_matrices, depth_values, tmp, position3d=None)`` takes and
returns exactly what the reference does (cost_volume.py:21-133) and owns the same parameters (``vis.*``,
``cost_reg.*``), but the work is seven HIP launches on the caller's stream instead of ~150 ATen ops:

    compose_homography -> warp_corr_entropy (all views) -> vis CNN (all views) -> warp_corr_aggregate
    -> regulariser U-Net (9 MFMA conv launches) -> prob + softmax + regression + confidence

With ``view_group`` set (a torch.distributed process group over RCCL) the source views are sharded over the
ranks of the group and the partial ``volume_sum`` / ``vis_sum`` are combined with ONE all-reduce per stage
(SURVEY.md section 8e); everything after the all-reduce is replicated, so all ranks return identical results.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, ops, packing
from .module import (DEFAULT_PRECISION, ConvBnReLU, CostRegNet, CostRegNet3D, PureTransformerCostReg, _bn_dict, _no_grad_path,
                     _PackedCache, precision_code)


def shard_views(n_src: int, world: int, rank: int):
    """Contiguous, balanced split of source views 1..n_src over `world` ranks -> [begin, end) (1-based view ids)."""
    base, extra = divmod(n_src, world)
    begin = 1 + rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class StageNet(nn.Module):
    def __init__(self, args: dict, ndepth: int, stage_idx: int):
        super().__init__()
        self.args = args
        self.fusion_type = args.get("fusion_type", "cnn")
        self.ndepth = ndepth
        self.stage_idx = stage_idx
        self.cost_reg_type = args.get("cost_reg_type", ["Normal"] * 4)[stage_idx]
        depth_type = args["depth_type"]
        self.depth_type = depth_type[stage_idx] if isinstance(depth_type, (list, tuple)) else depth_type
        ch = args["base_ch"]
        self.in_channels = ch[stage_idx] if isinstance(ch, (list, tuple)) else ch
        if self.fusion_type != "cnn":
            raise NotImplementedError(f"Not implemented fusion type: {self.fusion_type}.")
        self.vis = nn.Sequential(ConvBnReLU(1, 16), ConvBnReLU(16, 16), ConvBnReLU(16, 8), nn.Conv2d(8, 1, 1), nn.Sigmoid())
        if self.cost_reg_type == "PureTransformerCostReg":                                        # cost_volume.py:41-43
            args["transformer_config"][stage_idx]["base_channel"] = self.in_channels
            self.cost_reg = PureTransformerCostReg(self.in_channels, **args["transformer_config"][stage_idx])
        elif self.cost_reg_type != "Normal":
            raise NotImplementedError("cost_reg_type=%r" % self.cost_reg_type)
        elif ndepth <= args.get("model_th", 8):
            self.cost_reg = CostRegNet3D(self.in_channels, self.in_channels)
        else:
            self.cost_reg = CostRegNet(self.in_channels, self.in_channels)
        self.view_group = None            # torch.distributed group for view sharding (None = single GPU)
        self.return_prob_volumes = True   # prob_volume / prob_volume_pre are only read by the training losses
        # contraction of every MFMA convolution of the stage: "bf16x3" (3-term split bf16, ~2^-16 relative) or "fp32"
        self.conv_precision = args.get("conv_precision", DEFAULT_PRECISION)
        self._vis_cache = _PackedCache()

    # ---- packed parameters ----
    def _vis_params(self, device):
        prec = self.conv_precision
        pack = packing.pack_conv_weights_bf16x3 if prec == "bf16x3" else packing.pack_conv_weights
        precision_code(prec)

        def build(dev):
            out = []
            for i, ch in ((0, None), (1, 16), (2, 16)):
                layer = self.vis[i]
                w, b = packing.fold_bn(layer.conv.weight.detach().cpu().float(), _bn_dict(layer.bn), 0)
                if i == 0:
                    out += [w[:, 0].permute(1, 2, 0).reshape(9, 16).contiguous().to(dev), b.contiguous().to(dev)]
                else:
                    out += [pack(w.unsqueeze(2), 16).to(dev), packing.pad_bias(b).to(dev)]
            last = self.vis[3]
            out += [last.weight.detach().float().reshape(8).contiguous().to(dev), last.bias.detach().float().reshape(1).contiguous().to(dev)]
            return out
        return self._vis_cache.get(self.vis, build, prec)

    def forward(self, features, proj_matrices, depth_values, tmp, position3d=None) -> Dict[str, torch.Tensor]:
        _no_grad_path(features, depth_values)
        B, V, C, H, W = features.shape
        assert V == proj_matrices.shape[1], "Different number of images and projection matrices"   # cost_volume.py:56
        G = self.in_channels
        if G > C:
            raise AssertionError("G must <= C!")                                                  # cost_volume.py:87
        if G != 8:
            raise NotImplementedError("base_ch=%d: the HIP regulariser is built for 8 groups (all shipped configs)" % G)
        feats, code = ops._feat(features)
        hyp = ops._f32c(depth_values)
        if hyp.dim() != 4:
            raise ValueError("depth_values must be [B,D,H,W] inside the cascade")
        hom = ops.compose_homography(proj_matrices)
        vis_params = self._vis_params(feats.device)

        prec = precision_code(self.conv_precision)
        # pass 1 (entropy per view) -> visibility CNN -> pass 2 gathers again and writes the cost volume once: the per-view
        # correlation volumes are never kept (round 1 kept them for D >= 8: 2 x 32 B per voxel and view of HBM traffic)
        if self.view_group is None:
            entropy = ops.warp_corr_entropy(feats, code, hom, hyp, G)
            vis = ops.vis_weight(entropy, vis_params, prec)
            volume, _ = ops.warp_corr_aggregate(feats, code, hom, hyp, vis, G, normalise=True)
        else:
            volume = self._sharded_volume(feats, code, hom, hyp, G, vis_params)

        D = hyp.shape[1]
        conf_n = 0
        if self.depth_type == "ce":
            mode = _lib.HEAD_CE_TRAIN if self.training else _lib.HEAD_CE_EVAL
        else:
            mode = _lib.HEAD_REG
            conf_n = 4 if D >= 32 else (3 if D == 16 else (2 if D == 8 else 0))                    # cost_volume.py:121-128
        if isinstance(self.cost_reg, PureTransformerCostReg):
            prob_volume_pre = self.cost_reg.logits_cl(volume, position3d)
            depth, conf, prob_volume = ops.softmax_regress(prob_volume_pre, hyp, float(tmp), mode, conf_n, self.return_prob_volumes)
        else:
            ws, bs, prob_w, prob_b = self.cost_reg.packed_all(feats.device, self.conv_precision)
            feat_cl = ops.regnet(self.cost_reg.kind, volume, ws, bs, prec)
            depth, conf, prob_volume, prob_volume_pre = ops.prob_regress(
                feat_cl, prob_w, prob_b, self.cost_reg.prob_ksize, hyp, float(tmp), mode, conf_n, self.return_prob_volumes)
        return {"depth": depth, "prob_volume": prob_volume, "photometric_confidence": conf,
                "depth_values": depth_values, "prob_volume_pre": prob_volume_pre}

    # ---- source views sharded over ranks, one all-reduce(sum) of [G*D*HW + HW] floats ----
    def _sharded_volume(self, feats, code, hom, hyp, G, vis_params):
        import torch.distributed as dist
        B, V, C, H, W = feats.shape
        D = hyp.shape[1]
        world, rank = dist.get_world_size(self.view_group), dist.get_rank(self.view_group)
        vb, ve = shard_views(V - 1, world, rank)
        flat = torch.zeros(B * D * H * W * G + B * H * W, dtype=torch.float32, device=feats.device)
        vol = flat[: B * D * H * W * G].view(B, D, H, W, G)
        vsum = flat[B * D * H * W * G:].view(B, H, W)
        if ve > vb:
            entropy = ops.warp_corr_entropy(feats, code, hom, hyp, G, vb, ve)
            vis = entropy.clone()
            vis[:, vb - 1: ve - 1] = ops.vis_weight(entropy[:, vb - 1: ve - 1].contiguous(), vis_params, precision_code(self.conv_precision))
            ops.warp_corr_aggregate(feats, code, hom, hyp, vis, G, normalise=False, view_begin=vb, view_end=ve, out=(vol, vsum))
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.view_group)
        return ops.volume_normalise_(vol, vsum)
