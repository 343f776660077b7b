"""Comparator for the training-path tests and scripts (NOT part of the product): one cascade stage in train mode with every conv /
BatchNorm layer on PyTorch autograd ops - the reference's op graph (cost_volume.py:89-117, module.py:168-197, 393-408, 488-504) on the
module's own parameters - between the library's cost-volume construction (gather forward AND backward are the HIP kernels on both
routes) and the shared head.  `scripts/fuzz_train_gpu.py`, `scripts/prof_train.py` and `parity_cases.case_regnet_train_native`
compare the native route (mvsformerplusplus_amd.training) against this one."""
import torch
import torch.nn.functional as F
import torch.utils.checkpoint as cp

from mvsformerplusplus_amd import ops, training as T


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


def stage_forward_train_torch(net, features, proj_matrices, depth_values, tmp, position3d=None):
    """StageNet.forward in train mode with the conv / BatchNorm layers on autograd ops (the visibility CNN once per source view, as the
    reference calls it)."""
    from mvsformerplusplus_amd.module import PureTransformerCostReg
    B, V, C, H, W = features.shape
    G = net.in_channels
    with torch.no_grad():
        hyp = ops._f32c(depth_values)
        feats, code = ops._feat(features)
        hom = ops.compose_homography(proj_matrices)
        entropy = ops.warp_corr_entropy(feats, code, hom, hyp, G)
    vis = torch.cat([vis_forward_torch(net.vis, entropy[:, v:v + 1]) for v in range(V - 1)], dim=1)
    volume = T.WarpCorrAggregate.apply(features, vis, hom, hyp, G)                                # [B,G,D,H,W]
    if isinstance(net.cost_reg, PureTransformerCostReg):
        pre = T.transformer_forward_torch(net.cost_reg, volume, position3d).squeeze(1)
    else:
        pre = regnet_forward_torch(net.cost_reg, volume).squeeze(1)
    return T.stage_head_train(net, pre, depth_values, tmp)


def stage_forward_train_cpu(net, features, proj_matrices, depth_values, tmp):
    """The same stage ENTIRELY on PyTorch CPU autograd - no library kernel anywhere: warping / correlation / entropy from the oracle's
    differentiable restatement (oracle/ref_path.py: F.grid_sample under no_grad for the grid, entropy from sim.detach(), as the
    reference), the module's own Conv / BatchNorm layers in train mode, the shared autograd head.  `net` and all tensors on the CPU."""
    from oracle import ref_path as O
    B, V, C, H, W = features.shape
    G = net.in_channels
    ref_p = O.compose_proj(proj_matrices[:, 0])
    vol_sum, vis_sum = 0.0, 0.0
    for v in range(1, V):
        warped, _ = O.homo_warping_3D_with_mask(features[:, v], O.compose_proj(proj_matrices[:, v]), ref_p, depth_values)
        ip = O.group_correlation(features[:, 0], warped, G)                              # [B,G,D,H,W]
        ent = O.entropy_of_similarity(ip.detach())                                       # cost_volume.py:90-92
        vis = vis_forward_torch(net.vis, ent)                                            # [B,1,H,W]
        vol_sum = vol_sum + ip * vis.unsqueeze(1)
        vis_sum = vis_sum + vis
    volume = vol_sum / (vis_sum.unsqueeze(1) + 1e-6)
    pre = regnet_forward_torch(net.cost_reg, volume).squeeze(1)
    return T.stage_head_train(net, pre, depth_values, tmp)
