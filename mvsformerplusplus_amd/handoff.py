"""Producer side of the feature hand-off.

The reference's feature side ends, per stage, in a 3x3 ``Conv2d`` whose planar ``[B, C, H, W]`` results are stacked over the
views into ``features[stage] = [B, V, C, H, W]`` (``models/FMT.py:195-197, 231-240``: ``smooth_1/2/3``, no bias; or, for networks
that feed the FPN heads to the cost volume directly, ``models/module.py:257-270``: ``FPNDecoder.out1/2/3`` = Conv2d + BatchNorm2d +
Swish).  ``TiledFeatureHead`` wraps that LAST layer: same parameters (it holds the reference's own ``nn.Conv2d`` / ``nn.BatchNorm2d``
modules, so a checkpoint loads unchanged), but the forward runs ``mvs_conv2d3x3_tiles_fwd`` - split-bf16 MFMA convolution whose
epilogue writes the octet-tiled hand-off layout ``[B, V, C/8, H, W, 8]`` the gather passes read (``ops.PackedFeatures``): the planar
tensor, the ``torch.stack`` and the ``mvs_pack_features`` pass disappear.  INTEGRATION.md section 1b shows the three-line change
in the reference's ``FMT_with_pathway.forward``.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import ops, packing
from .module import _PackedCache, _bn_dict


class TiledFeatureHead(nn.Module):
    """conv: the reference's ``nn.Conv2d(Cin, Cout, 3, padding=1)``; bn: the ``nn.BatchNorm2d`` that follows it (eval mode, folded) or
    None; swish: apply ``x * sigmoid(x)`` (the FPN heads' ``Swish``).  (Cin, Cout) must be one of the reference's head shapes
    (``ops.feature_conv_is_built``)."""

    def __init__(self, conv: nn.Conv2d, bn: Optional[nn.BatchNorm2d] = None, swish: bool = False, dtype: torch.dtype = torch.bfloat16):
        super().__init__()
        if tuple(conv.kernel_size) != (3, 3) or tuple(conv.padding) != (1, 1) or tuple(conv.stride) != (1, 1) or tuple(conv.dilation) != (1, 1) or conv.groups != 1:
            raise NotImplementedError("TiledFeatureHead wraps Conv2d(k=3, padding=1, stride=1) - the reference's feature heads (FMT.py:195-197, module.py:257-270)")
        self.conv, self.bn, self.swish, self.dtype = conv, bn, bool(swish), dtype
        self._cache = _PackedCache()

    @staticmethod
    def from_sequential(seq: nn.Sequential, dtype: torch.dtype = torch.bfloat16) -> "TiledFeatureHead":
        """``FPNDecoder.outK`` = Sequential(Conv2d, BatchNorm2d, Swish) (module.py:247-256)."""
        return TiledFeatureHead(seq[0], seq[1], swish=len(seq) > 2, dtype=dtype)

    def _params(self, device):
        def build(dev):
            w = self.conv.weight.detach().cpu().float()
            b = self.conv.bias.detach().cpu().float() if self.conv.bias is not None else None
            if self.bn is not None:
                bnd = _bn_dict(self.bn)
                w, shift = packing.fold_bn(w, bnd, 0)
                if b is not None:                      # a conv bias in front of the BatchNorm goes through the same scale
                    scale = bnd["weight"].double() / torch.sqrt(bnd["running_var"].double() + bnd["eps"])
                    shift = (shift.double() + b.double() * scale).float()
                b = shift
            cin = w.shape[1]
            wp = packing.pack_conv_weights_bf16x3(w.unsqueeze(2), min(cin, 32)).to(dev)
            return wp, (b.contiguous().to(dev) if b is not None else None)
        mod = self if self.bn is not None else self.conv
        return self._cache.get(mod, build)

    def new_buffer(self, B: int, V: int, H: int, W: int, device) -> ops.PackedFeatures:
        """An empty hand-off tensor [B, V, Cout/8, H, W, 8] for `forward(..., out=buf, view=v)` to fill view by view."""
        return ops.PackedFeatures(torch.empty(B, V, self.conv.out_channels // 8, H, W, 8, dtype=self.dtype, device=device))

    def forward(self, x: torch.Tensor, out: Optional[ops.PackedFeatures] = None, view: Optional[int] = None) -> ops.PackedFeatures:
        """x [B, Cin, H, W] (one view, the reference's eval loop) with `out` + `view`: fills out[:, view] and returns `out`;
        x [B, V, Cin, H, W] or [B*V, Cin, H, W] without `out`: all views at once -> a new PackedFeatures [B, V | 1, Cout/8, H, W, 8]."""
        if (self.bn is not None and self.bn.training) or (torch.is_grad_enabled() and x.requires_grad):
            # inference-only (round 6): the kernel folds the BatchNorm's RUNNING statistics and has no backward - inside a model put
            # into train() it would silently use eval-mode BatchNorm and detach the feature side
            raise RuntimeError("TiledFeatureHead is an inference-time emitter (folded running BatchNorm statistics, no autograd): put the wrapped "
                               "BatchNorm2d into eval() and call it under torch.no_grad(), or keep the reference's Conv2d / BatchNorm2d / Swish + "
                               "ops.pack_features while training")
        with torch.no_grad():
            return self._forward(x, out, view)

    def _forward(self, x, out, view):
        wp, b = self._params(x.device)
        co = self.conv.out_channels
        if out is not None:
            assert view is not None and x.dim() == 4, "out=... takes one view [B,Cin,H,W] and its index"
            ops.conv2d3x3_tiles(x, wp, b, co, self.swish, out=out.data[:, view])
            return out
        if x.dim() == 5:
            B, V, cin, H, W = x.shape
            t = ops.conv2d3x3_tiles(x.reshape(B * V, cin, H, W), wp, b, co, self.swish, dtype=self.dtype)
            return ops.PackedFeatures(t.view(B, V, co // 8, H, W, 8))
        t = ops.conv2d3x3_tiles(x, wp, b, co, self.swish, dtype=self.dtype)
        return ops.PackedFeatures(t.unsqueeze(1))
