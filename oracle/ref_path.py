"""ORACLE (test infrastructure, NOT the product) - CPU restatement of MVSFormer++'s depth hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s baseline legs (``cpu_baseline``, and
``torch_rocm_composite`` = this same restatement with its tensors on cuda:0, both outside the timed
region) may import this module.  The product package ``mvsformerplusplus_amd`` never does: it fails loudly when its
HIP library is missing.

Every function restates, op for op and in fp32, one piece of the reference
(maybeLx/MVSFormerPlusPlus @ 2025-01-14) and cites the file:line it follows.  The arithmetic
that lives in a third-party dependency - PyTorch (``requirements.txt:9`` pins torch==2.1.2; the
container has 2.10.0): ``F.grid_sample``, ``conv3d``, ``conv_transpose3d``, ``batch_norm``,
``softmax``, ``interpolate(trilinear)`` - is called through the same public torch functions at the
reference's own call sites (``warping.py:105``, ``module.py:109-111,148-150,391,467-486,723``,
``cost_volume.py:91,106,115``).  An independent plain-C restatement of those ops from their
published definitions lives next to this file (``stage_ref.c``).

Parity pin: the reference has no tests or golden vectors of its own.  This oracle
is pinned against golden vectors produced by importing the reference itself in the build
container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``), checked by
``tests/test_oracle_golden.py``.

Weights are plain ``dict[str, Tensor]`` in the reference's state-dict naming (``vis.0.conv.weight``,
``cost_reg.conv1.bn.running_mean`` ...), so one ``.npz`` feeds the oracle, the HIP path and the
reference alike.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
BN_EPS = 1e-5  # nn.BatchNorm2d / nn.BatchNorm3d default, module.py:110,193


# --------------------------------------------------------------------------------------
# a1  projection compose                                        cost_volume.py:68-71
# --------------------------------------------------------------------------------------
def compose_proj(proj: torch.Tensor) -> torch.Tensor:
    """``proj[B,2,4,4]`` (0 = extrinsic, 1 = intrinsic) -> ``P[B,4,4]`` with ``P[:3,:4] = K[:3,:3] @ E[:3,:4]``."""
    out = proj[:, 0].clone()
    out[:, :3, :4] = torch.matmul(proj[:, 1, :3, :3], proj[:, 0, :3, :4])
    return out


# --------------------------------------------------------------------------------------
# a2/a3  homography warp                                         warping.py:69-109
# --------------------------------------------------------------------------------------
def homo_warping_3D_with_mask(src_fea: torch.Tensor, src_proj: torch.Tensor, ref_proj: torch.Tensor,
                              depth_values: torch.Tensor):
    """src_fea [B,C,H,W]; src_proj/ref_proj [B,4,4]; depth_values [B,D] or [B,D,H,W]
    -> (warped [B,C,D,H,W], proj_mask [B,D,H,W] bool)."""
    B, C, H, W = src_fea.shape
    D = depth_values.shape[1]
    proj = torch.matmul(src_proj, torch.inverse(ref_proj))                       # :80
    rot = proj[:, :3, :3]
    trans = proj[:, :3, 3:4]
    y, x = torch.meshgrid([torch.arange(0, H, dtype=torch.float32, device=src_fea.device),
                           torch.arange(0, W, dtype=torch.float32, device=src_fea.device)], indexing="ij")   # :84-85
    y, x = y.reshape(H * W), x.reshape(H * W)
    xyz = torch.stack((x, y, torch.ones_like(x)))                                 # :88
    xyz = xyz.unsqueeze(0).repeat(B, 1, 1)
    rot_xyz = torch.matmul(rot, xyz)                                              # :90
    rot_depth_xyz = rot_xyz.unsqueeze(2).repeat(1, 1, D, 1) * depth_values.reshape(B, 1, D, -1)   # :91
    proj_xyz = rot_depth_xyz + trans.reshape(B, 3, 1, 1)                           # :92
    proj_xy = proj_xyz[:, :2] / (proj_xyz[:, 2:3] + 1e-6)                          # :93
    xn = proj_xy[:, 0] / ((W - 1) / 2) - 1                                         # :94
    yn = proj_xy[:, 1] / ((H - 1) / 2) - 1                                         # :95
    grid = torch.stack((xn, yn), dim=3)                                            # :96
    X_mask = (xn > 1) | (xn < -1)                                                  # :99
    Y_mask = (yn > 1) | (yn < -1)                                                  # :100
    proj_mask = (X_mask | Y_mask).reshape(B, D, H, W)
    z = proj_xyz[:, 2:3].reshape(B, D, H, W)
    proj_mask = proj_mask | (z <= 0)                                               # :103
    warped = F.grid_sample(src_fea, grid.reshape(B, D * H, W, 2), mode="bilinear",
                           padding_mode="zeros", align_corners=True)               # :105-106
    return warped.reshape(B, C, D, H, W), proj_mask


# --------------------------------------------------------------------------------------
# a4  group-wise correlation                                     cost_volume.py:74-87
# --------------------------------------------------------------------------------------
def group_correlation(ref_feat: torch.Tensor, warped: torch.Tensor, G: int) -> torch.Tensor:
    B, C, D, H, W = warped.shape
    if G < C:
        wv = warped.reshape(B, G, C // G, D, H, W)
        rv = ref_feat.reshape(B, G, C // G, 1, H, W).to(torch.float32)
        return (rv * wv).mean(dim=2)                                               # :82
    if G == C:
        return ref_feat.reshape(B, G, 1, H, W).to(torch.float32) * warped          # :84-85
    raise AssertionError("G must <= C!")                                            # :87


# --------------------------------------------------------------------------------------
# a5  visibility weight                        cost_volume.py:89-93, module.py:168-197
# --------------------------------------------------------------------------------------
def _bn(x: torch.Tensor, sd: SD, prefix: str) -> torch.Tensor:
    return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                        sd[prefix + ".weight"], sd[prefix + ".bias"], False, 0.1, BN_EPS)


def vis_weight(entropy: torch.Tensor, sd: SD, prefix: str = "vis") -> torch.Tensor:
    x = entropy
    for i in range(3):                                                             # ConvBnReLU x3, cost_volume.py:36
        x = F.conv2d(x, sd["%s.%d.conv.weight" % (prefix, i)], None, stride=1, padding=1)
        x = F.relu(_bn(x, sd, "%s.%d.bn" % (prefix, i)))
    x = F.conv2d(x, sd[prefix + ".3.weight"], sd[prefix + ".3.bias"])             # nn.Conv2d(8,1,1)
    return torch.sigmoid(x)


def entropy_of_similarity(in_prod_vol: torch.Tensor) -> torch.Tensor:
    sim = in_prod_vol.sum(dim=1)                                                   # :90
    p = F.softmax(sim, dim=1)                                                      # :91
    return (-p * torch.log(p + 1e-7)).sum(dim=1, keepdim=True)                     # :92


# --------------------------------------------------------------------------------------
# a7-a9  3D U-Net regularisers                                  module.py:89-165,367-504
# --------------------------------------------------------------------------------------
def _conv3d_bn_relu(x, sd, prefix, stride):
    """``Conv3d`` wrapper module.py:89-126: conv (no bias) -> BN -> ReLU."""
    x = F.conv3d(x, sd[prefix + ".conv.weight"], None, stride=stride, padding=1)
    return F.relu(_bn(x, sd, prefix + ".bn"))


def _deconv3d_bn_relu(x, sd, wkey, bnprefix, stride, output_padding):
    x = F.conv_transpose3d(x, sd[wkey], None, stride=stride, padding=1, output_padding=output_padding)
    return F.relu(_bn(x, sd, bnprefix))


def _inner(conv0: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """``self.inner`` module.py:385-388 / 481-484: nn.Conv3d(in_channels, base_channels, 1, 1) (with bias) when the two differ, else
    nn.Identity (then the state dict has no ``inner.*`` keys)."""
    if (p + "inner.weight") in sd:
        return F.conv3d(conv0, sd[p + "inner.weight"], sd[p + "inner.bias"], stride=1, padding=0)
    return conv0


def cost_regnet(x: torch.Tensor, sd: SD, prefix: str = "cost_reg") -> torch.Tensor:
    """``CostRegNet.forward_once`` module.py:398-408, any in_channels / base_channels (the widths are the state dict's); without
    ``prob.weight`` in the state dict it is the ``last_layer=False`` form and the features come back (module.py:406-408)."""
    p = prefix + "."
    conv0 = x
    conv2 = _conv3d_bn_relu(_conv3d_bn_relu(conv0, sd, p + "conv1", 2), sd, p + "conv2", 1)
    conv4 = _conv3d_bn_relu(_conv3d_bn_relu(conv2, sd, p + "conv3", 2), sd, p + "conv4", 1)
    x = _conv3d_bn_relu(_conv3d_bn_relu(conv4, sd, p + "conv5", 2), sd, p + "conv6", 1)
    x = conv4 + _deconv3d_bn_relu(x, sd, p + "conv7.conv.weight", p + "conv7.bn", 2, 1)
    x = conv2 + _deconv3d_bn_relu(x, sd, p + "conv9.conv.weight", p + "conv9.bn", 2, 1)
    x = _inner(conv0, sd, p) + _deconv3d_bn_relu(x, sd, p + "conv11.conv.weight", p + "conv11.bn", 2, 1)
    if (p + "prob.weight") not in sd:
        return x
    return F.conv3d(x, sd[p + "prob.weight"], None, stride=1, padding=1)          # module.py:391 (3x3x3, no bias)


def cost_regnet3d(x: torch.Tensor, sd: SD, prefix: str = "cost_reg") -> torch.Tensor:
    """``CostRegNet3D.forward_once`` module.py:494-504: stride (1,2,2), raw Sequential deconvs, 1x1x1 prob + bias."""
    p = prefix + "."
    s = (1, 2, 2)
    op = (0, 1, 1)
    conv0 = x
    conv2 = _conv3d_bn_relu(_conv3d_bn_relu(conv0, sd, p + "conv1", s), sd, p + "conv2", 1)
    conv4 = _conv3d_bn_relu(_conv3d_bn_relu(conv2, sd, p + "conv3", s), sd, p + "conv4", 1)
    x = _conv3d_bn_relu(_conv3d_bn_relu(conv4, sd, p + "conv5", s), sd, p + "conv6", 1)
    x = conv4 + _deconv3d_bn_relu(x, sd, p + "conv7.0.weight", p + "conv7.1", s, op)
    x = conv2 + _deconv3d_bn_relu(x, sd, p + "conv9.0.weight", p + "conv9.1", s, op)
    x = _inner(conv0, sd, p) + _deconv3d_bn_relu(x, sd, p + "conv11.0.weight", p + "conv11.1", s, op)
    return F.conv3d(x, sd[p + "prob.weight"], sd[p + "prob.bias"], stride=1, padding=0)    # module.py:486


def cost_regnet2d(x: torch.Tensor, sd: SD, prefix: str = "cost_reg") -> torch.Tensor:
    """``CostRegNet2D.forward`` module.py:439-450: CostRegNet3D's topology, (1,3,3) kernels with padding (0,1,1) in the strided and the
    transposed layers (conv2 / conv4 / conv6 stay 3x3x3), ``conv0`` added as it is (no ``inner``), 1x1x1 prob + bias."""
    p = prefix + "."
    s, pad, op = (1, 2, 2), (0, 1, 1), (0, 1, 1)

    def down(t, name):
        t = F.conv3d(t, sd[p + name + ".conv.weight"], None, stride=s, padding=pad)
        return F.relu(_bn(t, sd, p + name + ".bn"))

    def up(t, name):
        t = F.conv_transpose3d(t, sd[p + name + ".0.weight"], None, stride=s, padding=pad, output_padding=op)
        return F.relu(_bn(t, sd, p + name + ".1"))
    conv0 = x
    conv2 = _conv3d_bn_relu(down(conv0, "conv1"), sd, p + "conv2", 1)
    conv4 = _conv3d_bn_relu(down(conv2, "conv3"), sd, p + "conv4", 1)
    x = _conv3d_bn_relu(down(conv4, "conv5"), sd, p + "conv6", 1)
    x = conv4 + up(x, "conv7")
    x = conv2 + up(x, "conv9")
    x = conv0 + up(x, "conv11")
    return F.conv3d(x, sd[p + "prob.weight"], sd[p + "prob.bias"], stride=1, padding=0)


def is_regnet3d(sd: SD, prefix: str = "cost_reg") -> bool:
    return (prefix + ".conv7.0.weight") in sd


# --------------------------------------------------------------------------------------
# a10/a11  regression + confidence                                module.py:649-671
# --------------------------------------------------------------------------------------
def depth_regression(p: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    if depth_values.dim() <= 2:
        depth_values = depth_values.reshape(*depth_values.shape, 1, 1)
    return torch.sum(p * depth_values, 1)


def conf_regression(p: torch.Tensor, n: int = 4) -> torch.Tensor:
    ndepths = p.size(1)
    if n % 2 == 1:
        pad = [0, 0, 0, 0, n // 2, n // 2]
    else:
        pad = [0, 0, 0, 0, n // 2 - 1, n // 2]
    s = n * F.avg_pool3d(F.pad(p.unsqueeze(1), pad=pad), (n, 1, 1), stride=1, padding=0).squeeze(1)
    idx = depth_regression(p, torch.arange(ndepths, dtype=torch.float, device=p.device)).long().clamp(min=0, max=ndepths - 1)
    return torch.gather(s, 1, idx.unsqueeze(1)).squeeze(1)


# --------------------------------------------------------------------------------------
# a13-a15  hypothesis ranges                                     module.py:674-741
# --------------------------------------------------------------------------------------
def init_range(cur_depth, ndepths, H, W):
    dtype = cur_depth.dtype
    if cur_depth.dim() == 2:
        dmin, dmax = cur_depth[:, 0], cur_depth[:, -1]
        itv = ((dmax - dmin) / (ndepths - 1))[:, None, None]
        s = dmin.unsqueeze(1) + torch.arange(0, ndepths, dtype=dtype, device=cur_depth.device).reshape(1, -1) * itv.squeeze(1)
        return s.unsqueeze(-1).unsqueeze(-1).repeat(1, 1, H, W)
    dmin, dmax = cur_depth[..., 0], cur_depth[..., -1]
    itv = (dmax - dmin) / (ndepths - 1)
    return dmin.unsqueeze(1) + torch.arange(0, ndepths, dtype=dtype, device=cur_depth.device).reshape(1, -1, 1, 1) * itv.unsqueeze(1)


def init_inverse_range(cur_depth, ndepths, H, W):
    dtype = cur_depth.dtype
    itv = torch.arange(0, ndepths, dtype=dtype, device=cur_depth.device).reshape(1, -1, 1, 1).repeat(1, 1, H, W) / (ndepths - 1)
    if cur_depth.dim() == 2:
        inv_min = 1.0 / cur_depth[:, 0]
        inv_max = 1.0 / cur_depth[:, -1]
        hypo = inv_max[:, None, None, None] + (inv_min - inv_max)[:, None, None, None] * itv
    else:
        inv_min = 1.0 / cur_depth[..., 0]
        inv_max = 1.0 / cur_depth[..., -1]
        hypo = inv_max[:, None, :, :] + (inv_min - inv_max)[:, None, :, :] * itv
    return 1.0 / hypo


def schedule_inverse_range(depth, depth_hypo, ndepths, split_itv, H, W, shift=False):
    last_itv = 1.0 / depth_hypo[:, 2] - 1.0 / depth_hypo[:, 1]                      # :708
    inv_min = 1 / depth + split_itv * last_itv
    inv_max = 1 / depth - split_itv * last_itv
    if shift:                                                                       # :712-715 (unused by shipped configs)
        is_neg = (inv_max < 0.002).float()
        inv_max = inv_max - (inv_max - 0.002) * is_neg
        inv_min = inv_min - (inv_max - 0.002) * is_neg
    itv = torch.arange(0, ndepths, dtype=inv_min.dtype, device=inv_min.device).reshape(1, -1, 1, 1).repeat(1, 1, H // 2, W // 2) / (ndepths - 1)
    hypo = inv_max[:, None] + (inv_min - inv_max)[:, None] * itv
    hypo = F.interpolate(hypo.unsqueeze(1), [ndepths, H, W], mode="trilinear", align_corners=True).squeeze(1)   # :723
    return 1.0 / hypo


def schedule_range(cur_depth, ndepth, depth_interval_pixel, H, W):
    if not torch.is_tensor(depth_interval_pixel):
        depth_interval_pixel = torch.tensor(depth_interval_pixel, dtype=cur_depth.dtype, device=cur_depth.device)
    if depth_interval_pixel.dim() != 3:
        depth_interval_pixel = depth_interval_pixel.reshape(-1)[:, None, None]
    dmin = torch.clamp_min(cur_depth - ndepth / 2 * depth_interval_pixel, 0.001)
    dmax = cur_depth + ndepth / 2 * depth_interval_pixel
    itv = (dmax - dmin) / (ndepth - 1)
    s = dmin.unsqueeze(1) + torch.arange(0, ndepth, dtype=cur_depth.dtype, device=cur_depth.device).reshape(1, -1, 1, 1) * itv.unsqueeze(1)
    return F.interpolate(s.unsqueeze(1), [ndepth, H, W], mode="trilinear", align_corners=True).squeeze(1)


# --------------------------------------------------------------------------------------
# stage-1 transformer regulariser
#   PureTransformerCostReg module.py:602-646, FlashAttnBlock :535-583, FFN :507-532, LayerNorm3D :586-599,
#   attention models/dino/layers/attention.py:76-101,141-170, Frustoconical PE position_encoding.py:138-189
# --------------------------------------------------------------------------------------
def get_position_3d(H: int, W: int, K: torch.Tensor, depth_values: torch.Tensor, depth_min, depth_max,
                    height_min=None, height_max=None, width_min=None, width_max=None):
    """position_encoding.py:138-163 (normalize=True).  K [B,3,3]; depth_values [B,D,H,W] -> ([B,3,D,H,W], 4 range scalars)."""
    B, D = depth_values.shape[:2]
    dev = depth_values.device
    y, x = torch.meshgrid([torch.arange(0, H, dtype=torch.float32, device=dev), torch.arange(0, W, dtype=torch.float32, device=dev)], indexing="ij")
    xyz = torch.stack((x.reshape(-1), y.reshape(-1), torch.ones(H * W, device=dev)))[None].repeat(B, 1, 1)
    xyz = torch.matmul(torch.inverse(K), xyz)                                               # :147
    pos = xyz.unsqueeze(2).repeat(1, 1, D, 1) * depth_values.reshape(B, 1, D, -1)            # :149
    if height_min is None or height_max is None or width_min is None or width_max is None:
        width_min, width_max = pos[:, 0].min(), pos[:, 0].max()
        height_min, height_max = pos[:, 1].min(), pos[:, 1].max()
    pos[:, 0] = (pos[:, 0] - width_min) / (width_max - width_min + 1e-5)
    pos[:, 1] = (pos[:, 1] - height_min) / (height_max - height_min + 1e-5)
    pos[:, 2] = (torch.clamp(pos[:, 2], depth_min, depth_max) - depth_min) / (depth_max - depth_min + 1e-5)
    return pos.reshape(B, 3, D, H, W), height_min, height_max, width_min, width_max


def position_encoding_3d(position3d: torch.Tensor, C: int, rescale: float = 4.0) -> torch.Tensor:
    """position_encoding.py:166-189: per axis C channels sin/cos interleaved -> [B,3C,D,H,W]."""
    import math
    B, _, D, H, W = position3d.shape
    div = torch.exp(torch.arange(0, C, 2, device=position3d.device).float() * (-math.log(10000.0) / C))[None, :, None]
    parts = []
    for a in range(3):
        pe = torch.zeros(B, C, D * H * W, dtype=torch.float32, device=position3d.device)
        pos = position3d[:, a].reshape(B, 1, D * H * W)
        pe[:, 0::2] = torch.sin(pos * rescale * div)
        pe[:, 1::2] = torch.cos(pos * rescale * div)
        parts.append(pe)
    return torch.cat(parts, 1).reshape(B, 3 * C, D, H, W)


def layer_norm_3d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    u = x.mean(1, keepdim=True)                                                               # module.py:594-598
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return w[:, None, None, None] * x + b[:, None, None, None]


def is_transformer(sd: SD, prefix: str = "cost_reg.") -> bool:
    return (prefix + "down.0.weight") in sd


def pure_transformer_cost_reg(x: torch.Tensor, position3d, sd: SD, *, num_heads: int, train_avg_length,
                              softmax_scale="entropy_invariance", prefix: str = "cost_reg.") -> torch.Tensor:
    """PureTransformerCostReg.forward (post-norm blocks, softmax attention), module.py:629-646."""
    import math
    g = lambda k: sd[prefix + k]
    if position3d is not None:
        x = x + F.conv3d(position_encoding_3d(position3d, x.shape[1]), g("pe_proj.weight"))  # :633 (use_pe_proj)
    rate = tuple(g("down.0.weight").shape[2:])
    x = F.conv3d(x, g("down.0.weight"), g("down.0.bias"), stride=rate)
    x = layer_norm_3d(x, g("down.1.weight"), g("down.1.bias"))
    B, C, d, h, w = x.shape
    hd = C // num_heads
    n_layers = 1 + max(int(k[len(prefix):].split(".")[1]) for k in sd if k.startswith(prefix + "attention_layers."))
    for i in range(n_layers):
        L = lambda k: g("attention_layers.%d.%s" % (i, k))
        t = x.permute(0, 3, 4, 2, 1).reshape(B, h * w * d, C)                                 # "b c d h w -> b (h w d) c"  :573
        N = t.shape[1]
        qkv = F.linear(t, L("attn.qkv.weight")).reshape(B, N, 3, num_heads, hd).permute(2, 0, 3, 1, 4)
        scale = hd ** -0.5
        if softmax_scale == "entropy_invariance":
            scale = scale * math.log(N, train_avg_length)                                     # attention.py:82-83,161
        att = F.scaled_dot_product_attention(qkv[0], qkv[1], qkv[2], scale=scale)            # attention.py:96 (no flash-attn on CPU)
        a = F.linear(att.transpose(1, 2).reshape(B, N, C), L("attn.proj.weight"), L("attn.proj.bias"))
        t = F.layer_norm(t + L("gamma1") * a, (C,), L("norm1.weight"), L("norm1.bias"), 1e-5)           # :575
        f = F.linear(F.gelu(F.linear(t, L("ffn.linear1.weight"), L("ffn.linear1.bias"))), L("ffn.linear2.weight"), L("ffn.linear2.bias"))
        t = F.layer_norm(t + L("gamma2") * f, (C,), L("norm2.weight"), L("norm2.bias"), 1e-5)           # :576
        x = t.reshape(B, h, w, d, C).permute(0, 4, 3, 1, 2)
    x = F.conv_transpose3d(x, g("up.0.weight"), g("up.0.bias"), stride=rate)
    x = layer_norm_3d(x, g("up.1.weight"), g("up.1.bias"))
    return F.conv3d(x, g("prob.weight"), g("prob.bias"))


# --------------------------------------------------------------------------------------
# SURVEY section 8f #4 (producer side of the feature hand-off): the feature side's last 3x3 convolutions
# --------------------------------------------------------------------------------------
def feature_head(x: torch.Tensor, weight: torch.Tensor, bias=None, bn=None, swish: bool = False) -> torch.Tensor:
    """FMT_with_pathway.smooth_k (models/FMT.py:195-197: Conv2d(C, C, 3, padding=1, bias=False), applied per view at :231-233) and
    FPNDecoder.out_k (models/module.py:247-256: Conv2d(64, C, 3, padding=1) -> BatchNorm2d -> Swish, applied at :262-268).
    x [N,Cin,H,W] fp32; bn = dict(weight, bias, running_mean, running_var, eps) of the eval-mode BatchNorm2d or None."""
    y = F.conv2d(x, weight, bias, padding=1)
    if bn is not None:
        y = F.batch_norm(y, bn["running_mean"], bn["running_var"], bn["weight"], bn["bias"], False, 0.0, float(bn["eps"]))
    if swish:
        y = y * torch.sigmoid(y)                                                    # module.py Swish
    return y


# --------------------------------------------------------------------------------------
# StageNet.forward                                               cost_volume.py:51-133
# --------------------------------------------------------------------------------------
def stage_forward(features: torch.Tensor, proj_matrices: torch.Tensor, depth_values: torch.Tensor, tmp: float,
                  sd: SD, *, G: int, depth_type: str = "ce", training: bool = False,
                  return_intermediates: bool = False, position3d=None, transformer_config=None) -> Dict[str, torch.Tensor]:
    """features [B,V,C,H,W]; proj_matrices [B,V,2,4,4]; depth_values [B,D,H,W]."""
    ref_feat = features[:, 0]
    V = features.shape[1]
    ndepth = depth_values.shape[1]
    assert proj_matrices.shape[1] == V, "Different number of images and projection matrices"
    ref_proj = compose_proj(proj_matrices[:, 0])
    volume_sum = 0.0
    vis_sum = 0.0
    entropies, vises = [], []
    for v in range(1, V):
        src_feat = features[:, v].to(torch.float32)                                # :67
        src_proj = compose_proj(proj_matrices[:, v])
        warped, _ = homo_warping_3D_with_mask(src_feat, src_proj, ref_proj, depth_values)
        in_prod = group_correlation(ref_feat, warped, G)
        ent = entropy_of_similarity(in_prod)
        w = vis_weight(ent, sd)
        volume_sum = volume_sum + in_prod * w.unsqueeze(1)                         # :97
        vis_sum = vis_sum + w                                                      # :98
        entropies.append(ent)
        vises.append(w)
    volume_mean = volume_sum / (vis_sum.unsqueeze(1) + 1e-6)                       # :101
    if is_transformer(sd):
        tc = transformer_config or {}
        cost = pure_transformer_cost_reg(volume_mean, position3d, sd, num_heads=tc.get("num_heads", 8),
                                         train_avg_length=tc.get("train_avg_length"), softmax_scale=tc.get("softmax_scale"))
    else:
        cost = cost_regnet3d(volume_mean, sd) if is_regnet3d(sd) else cost_regnet(volume_mean, sd)   # :103
    pre = cost.squeeze(1)
    prob = F.softmax(pre, dim=1)                                                   # :106
    if depth_type == "ce":
        if training:
            idx = torch.max(prob, dim=1)[1]
            depth = torch.gather(depth_values, 1, idx.unsqueeze(1)).squeeze(1)     # :109-112
        else:
            depth = depth_regression(F.softmax(pre * tmp, dim=1), depth_values)    # :115
        conf = prob.max(1)[0]                                                      # :117
    else:
        depth = depth_regression(prob, depth_values)                               # :120
        if ndepth >= 32:
            conf = conf_regression(prob, n=4)
        elif ndepth == 16:
            conf = conf_regression(prob, n=3)
        elif ndepth == 8:
            conf = conf_regression(prob, n=2)
        else:
            conf = prob.max(1)[0]
    out = {"depth": depth, "prob_volume": prob, "photometric_confidence": conf,
           "depth_values": depth_values, "prob_volume_pre": pre}
    if return_intermediates:
        out["volume_mean"] = volume_mean
        out["entropy"] = torch.stack(entropies, 1)
        out["vis_weight"] = torch.stack(vises, 1)
    return out


# --------------------------------------------------------------------------------------
# a16  cascade driver                         DINOv2_mvsformer_model.py:120-179
# --------------------------------------------------------------------------------------
def cascade_forward(features: Dict[str, torch.Tensor], proj_matrices: Dict[str, torch.Tensor],
                    depth_values: torch.Tensor, sds: Sequence[SD], *, ndepths: Sequence[int],
                    depth_interals_ratio: Sequence[float], base_ch: Sequence[int],
                    tmp: Sequence[float] = (5.0, 5.0, 5.0, 1.0), inverse_depth: bool = True,
                    depth_type: Sequence[str] = ("ce", "ce", "ce", "ce"), training: bool = False,
                    full_hw=None, use_pe3d: bool = False, transformer_config=None) -> Dict[str, torch.Tensor]:
    """Cascade from per-stage features (no image backbone); a stage whose state dict holds a transformer regulariser
    gets the Frustoconical position encoding when ``use_pe3d`` (DINOv2_mvsformer_model.py:151-162)."""
    n = len(ndepths)
    f_last = features["stage%d" % n]
    B = f_last.shape[0]
    Hf, Wf = full_hw if full_hw is not None else f_last.shape[-2:]
    depth_interval = depth_values[:, 1] - depth_values[:, 0]
    prob_maps = torch.zeros(B, Hf, Wf, dtype=torch.float32, device=f_last.device)
    outputs: Dict[str, torch.Tensor] = {}
    st = None
    pe_range = [None, None, None, None]                      # height_min, height_max, width_min, width_max  (:154-160)
    for s in range(n):
        proj = proj_matrices["stage%d" % (s + 1)]
        feat = features["stage%d" % (s + 1)]
        H, W = feat.shape[-2:]
        if s == 0:
            hyp = init_inverse_range(depth_values, ndepths[s], H, W) if inverse_depth else init_range(depth_values, ndepths[s], H, W)
        elif inverse_depth:
            hyp = schedule_inverse_range(st["depth"], st["depth_values"], ndepths[s], depth_interals_ratio[s], H, W)
        else:
            hyp = schedule_range(st["depth"], ndepths[s], depth_interals_ratio[s] * depth_interval, H, W)
        position3d = None
        if is_transformer(sds[s]) and use_pe3d:
            position3d, *pe_range = get_position_3d(H, W, proj[:, 0, 1, :3, :3], hyp, depth_values.min(), depth_values.max(), *pe_range)
        st = stage_forward(feat, proj, hyp, tmp[s], sds[s], G=base_ch[s], depth_type=depth_type[s], training=training,
                           position3d=position3d, transformer_config=(transformer_config or [None] * n)[min(s, len(transformer_config or [None] * n) - 1)])
        outputs["stage%d" % (s + 1)] = st
        conf = st["photometric_confidence"]
        if conf.shape[1] != Hf or conf.shape[2] != Wf:
            conf = F.interpolate(conf.unsqueeze(1), [Hf, Wf], mode="nearest").squeeze(1)   # :168-170
        prob_maps = prob_maps + conf
        outputs.update(st)
    outputs["refined_depth"] = st["depth"]
    outputs["photometric_confidence"] = prob_maps / n
    return outputs
