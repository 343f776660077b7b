"""Training-mode path of one cascade stage (first slice).

What is native and what is not, stated plainly:

* the cost-volume construction - homography compose, warp + group-wise correlation + entropy (pass 1), the visibility-weighted
  aggregation (pass 2) AND ITS BACKWARD - runs on the library's HIP kernels: ``WarpCorrAggregate`` is a
  ``torch.autograd.Function`` whose forward is ``mvs_warp_corr_aggregate_fwd`` and whose backward is
  ``mvs_warp_corr_aggregate_bwd`` (gradients w.r.t. reference features, source features and visibility maps; the warped
  [B,C,D,H,W] volumes the reference keeps alive per view for ``grid_sample``'s backward are never materialised);
* the visibility CNN, the 3-D U-Net regulariser (under ``torch.utils.checkpoint`` like the reference, module.py:393-396 /
  488-492) and the softmax / regression head run as PyTorch-ROCm autograd ops on the module's own ``nn.Conv*`` /
  ``nn.BatchNorm*`` parameters, so batch statistics, running-stat updates, SyncBatchNorm conversion and DDP (train.py:196-200)
  behave exactly as in the reference.  Hand-written conv / BatchNorm backward kernels are NOT part of this slice.

The reference differentiates neither the sampling grid (built under ``torch.no_grad()``, warping.py:80) nor the entropy
(``sim_vol.detach()``, cost_volume.py:90); this path follows it: no gradient reaches the depth hypotheses, the cameras or - via
the entropy - the features.  The transformer regulariser of the shipped stage 1 has no training path here and raises.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from . import _lib, ops


class WarpCorrAggregate(torch.autograd.Function):
    """volume_mean [B,G,D,H,W] = sum_v in_prod_v * vis_v / (sum_v vis_v + 1e-6)   (cost_volume.py:74-101), HIP forward and backward."""

    @staticmethod
    def forward(ctx, features, vis, homography, hyp, G):
        feats, code = ops._feat(features.detach())
        vis_c = ops._f32c(vis.detach())
        vol_cl, _ = ops.warp_corr_aggregate(feats, code, homography, hyp, vis_c, G, normalise=True)
        ctx.save_for_backward(feats, vis_c, homography, hyp, vol_cl)
        ctx.code, ctx.G, ctx.in_dtype = code, G, features.dtype
        return ops.cl_to_ncdhw(vol_cl)

    @staticmethod
    def backward(ctx, grad_volume):
        feats, vis, hom, hyp, vol_cl = ctx.saved_tensors
        gvol_cl = ops.ncdhw_to_cl(ops._f32c(grad_volume))
        vis_sum = vis.sum(dim=1).contiguous()
        gfeat, gvis = ops.warp_corr_aggregate_bwd(feats, ctx.code, hom, hyp, vis, vis_sum, vol_cl, gvol_cl, ctx.G)
        return gfeat.to(ctx.in_dtype), gvis, None, None, None


def regnet_forward_torch(reg, x: torch.Tensor) -> torch.Tensor:
    """CostRegNet / CostRegNet3D forward (module.py:398-408, 494-504) as autograd ops on the module's own layers."""
    def block(layer, t):
        if isinstance(layer, torch.nn.Sequential):                 # CostRegNet3D's conv7/9/11: ConvTranspose3d, BatchNorm3d, ReLU
            return layer(t)
        t = layer.conv(t)
        if layer.bn is not None:
            t = layer.bn(t)
        return F.relu(t) if layer.relu else t

    def once(v):
        conv0 = v
        conv2 = block(reg.conv2, block(reg.conv1, conv0))
        conv4 = block(reg.conv4, block(reg.conv3, conv2))
        t = block(reg.conv6, block(reg.conv5, conv4))
        t = conv4 + block(reg.conv7, t)
        t = conv2 + block(reg.conv9, t)
        t = reg.inner(conv0) + block(reg.conv11, t)
        return reg.prob(t)
    if torch.is_grad_enabled() and x.requires_grad:
        return cp.checkpoint(once, x, use_reentrant=True)
    return once(x)


def vis_forward_torch(vis_seq, entropy: torch.Tensor) -> torch.Tensor:
    """self.vis(entropy) (cost_volume.py:36,93): three Conv2d + BatchNorm2d + ReLU, a 1x1 Conv2d and a sigmoid.
    entropy [B,1,H,W] -> [B,1,H,W]."""
    t = entropy
    for i in range(3):
        t = F.relu(vis_seq[i].bn(vis_seq[i].conv(t)))
    return torch.sigmoid(vis_seq[3](t))


def stage_forward_train(net, features, proj_matrices, depth_values, tmp) -> Dict[str, torch.Tensor]:
    """StageNet.forward with autograd (cost_volume.py:51-133).  See the module docstring for what runs where."""
    from .module import PureTransformerCostReg
    if isinstance(net.cost_reg, PureTransformerCostReg):
        raise NotImplementedError("training through the transformer regulariser is not implemented (SURVEY.md section 8f #2 covers the "
                                  "'Normal' regularisers in this slice)")
    if isinstance(features, ops.PackedFeatures):
        raise NotImplementedError("the training path takes planar [B,V,C,H,W] features")
    if net.view_group is not None:
        raise NotImplementedError("view sharding is an inference-latency mode; train with DistributedDataParallel over batches (train.py:196-200)")
    B, V, C, H, W = features.shape
    G = net.in_channels
    if G > C:
        raise AssertionError("G must <= C!")                                                      # cost_volume.py:87
    with torch.no_grad():
        hyp = ops._f32c(depth_values)
        feats, code = ops._feat(features)
        hom = ops.compose_homography(proj_matrices)
        entropy = ops.warp_corr_entropy(feats, code, hom, hyp, G)                                 # [B,V-1,H,W], from sim.detach() in the reference
    # the reference runs the visibility CNN once per source view on a batch of B maps (cost_volume.py:93); BatchNorm statistics
    # are per call there, so the views are kept as separate calls here as well
    vis = torch.cat([vis_forward_torch(net.vis, entropy[:, v:v + 1]) for v in range(V - 1)], dim=1)        # [B,V-1,H,W]
    volume = WarpCorrAggregate.apply(features, vis, hom, hyp, G)                                   # [B,G,D,H,W]
    prob_volume_pre = regnet_forward_torch(net.cost_reg, volume).squeeze(1)
    prob_volume = F.softmax(prob_volume_pre, dim=1)
    D = hyp.shape[1]
    if net.depth_type == "ce":
        if net.training:
            idx = prob_volume.argmax(dim=1, keepdim=True)
            depth = torch.gather(depth_values, 1, idx).squeeze(1)                                 # cost_volume.py:109-112
        else:
            depth = (F.softmax(prob_volume_pre * tmp, dim=1) * depth_values).sum(1)
        conf = prob_volume.max(1)[0]
    else:
        depth = (prob_volume * depth_values).sum(1)
        n = 4 if D >= 32 else (3 if D == 16 else (2 if D == 8 else 0))                            # cost_volume.py:121-128
        conf = conf_regression_torch(prob_volume, n) if n else prob_volume.max(1)[0]
    return {"depth": depth, "prob_volume": prob_volume, "photometric_confidence": conf.detach(),
            "depth_values": depth_values, "prob_volume_pre": prob_volume_pre}


def conf_regression_torch(p: torch.Tensor, n: int) -> torch.Tensor:
    """Confidence of module.py:658-671: the probability mass of the n planes around floor(E[plane index]) - the window is
    [i - n//2, i + n//2] for odd n and [i - n//2 + 1, i + n//2] for even n, clipped to the volume.  Detached like the reference's."""
    D = p.shape[1]
    with torch.no_grad():
        q = p.detach()
        planes = torch.arange(D, device=p.device, dtype=torch.float32).view(1, D, 1, 1)
        idx = (q * planes).sum(1, keepdim=True).long().clamp(0, D - 1)
        lo = n // 2 if n % 2 == 1 else n // 2 - 1
        csum = F.pad(q.cumsum(1), (0, 0, 0, 0, 1, 0))                                             # csum[k] = sum of planes < k
        first = (idx - lo).clamp(0, D)
        last = (idx + n // 2 + 1).clamp(0, D)
        return (torch.gather(csum, 1, last) - torch.gather(csum, 1, first)).squeeze(1)
