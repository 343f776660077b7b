"""Training-mode path of one cascade stage.

What is native and what is not, stated plainly:

* the cost-volume construction - homography compose, warp + group-wise correlation + entropy (pass 1), the visibility-weighted
  aggregation (pass 2) AND ITS BACKWARD - runs on the library's HIP kernels: ``WarpCorrAggregate`` is a
  ``torch.autograd.Function`` whose forward is ``mvs_warp_corr_aggregate_fwd`` and whose backward is
  ``mvs_warp_corr_aggregate_bwd`` (gradients w.r.t. reference features, source features and visibility maps; the warped
  [B,C,D,H,W] volumes the reference keeps alive per view for ``grid_sample``'s backward are never materialised);
* the 3-D U-Net regulariser (CostRegNet / CostRegNet3D, nine Conv3d / ConvTranspose3d + BatchNorm3d + ReLU blocks and three skip
  adds) runs on the library's kernels in both directions (``RegNetTrain``): forward convolutions and DATA gradients on the
  split-bf16 MFMA kernels of the inference path with un-folded, re-packed weights (the data gradient of a stride-1 convolution is
  the convolution with flipped, transposed taps; of a strided convolution the transposed convolution; of a transposed convolution
  the strided convolution), WEIGHT gradients on an fp32-MFMA kernel (``mvs_conv3d_wgrad``), batch-statistics BatchNorm + ReLU +
  skip forward and backward on ``mvs_bn_*`` (sums in double, SyncBatchNorm's all-reduce of the sums included); on one rank each
  block is ONE C call per direction (``mvs_train_block_fwd`` / ``_bwd`` chain pack, convolution, statistics, finalize, normalise /
  reduce, apply, weight gradient, pack, data gradient on the stream).  Activations (pre-BatchNorm convolution outputs and
  block outputs) are kept while small and RECOMPUTED in the backward pass above 512 MB, or when ``reg.recompute_in_backward`` says
  so - the reference always runs its regularisers under ``torch.utils.checkpoint`` (module.py:393-396, 488-492);
* the visibility CNN's three Conv2d + BatchNorm2d + ReLU blocks (``VisTrain``: the same kernels on D = 1 volumes, BatchNorm per
  source view like the reference's per-view calls) and CostRegNet's 3x3x3 `prob` (``Prob3Train``) are native as well;
* still PyTorch-ROCm autograd, all of it element-wise or tiny: the visibility CNN's 1x1 Conv2d + sigmoid, CostRegNet3D's 1x1x1 `prob`,
  the softmax / argmax / regression head, the running-statistics momentum update.  There is no second backend in this module: a
  stage the native kernels do not cover (base_ch != 8) raises; a head configured with conv_precision "fp32" (exact contraction at
  inference) trains on the fp32-equivalent split-bf16 kernels and says so once.  The comparator that routes every conv /
  BatchNorm layer through PyTorch-ROCm autograd ops lives with the tests (``tests/train_torch_route.py``; on the MI355X image MIOpen
  picks naive kernels for these 3-D and 2-D convolutions: 400 ms per stage-4 step against 11.8 ms natively).

The reference differentiates neither the sampling grid (built under ``torch.no_grad()``, warping.py:80) nor the entropy
(``sim_vol.detach()``, cost_volume.py:90); this path follows it: no gradient reaches the depth hypotheses, the cameras or - via
the entropy - the features.  The transformer regulariser of the shipped stage 1 (``transformer_forward_torch``): each of its six
blocks is one autograd node (``TransformerBlockTrain``) whose forward is the inference path's five launches and whose backward is
hand-written - the attention core on ``mvs_tr_attention_bwd`` (fp32, no [n, n] tensor), the linears' gradients as plain GEMMs
(rocBLAS), LayerNorm / GELU / residual-scale gradients as tensor expressions, the FFN hidden layer and the pre-LayerNorm sums
recomputed; patch embedding / expansion with their LayerNorm3D and `prob` stay PyTorch-ROCm autograd ops.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops, packing


class WarpCorrAggregate(torch.autograd.Function):
    """volume_mean [B,G,D,H,W] = sum_v in_prod_v * vis_v / (sum_v vis_v + 1e-6)   (cost_volume.py:74-101), HIP forward and backward."""

    @staticmethod
    def forward(ctx, features, vis, homography, hyp, G, channel_last=False):
        feats, code = ops._feat(features.detach())
        vis_c = ops._f32c(vis.detach())
        vol_cl, _ = ops.warp_corr_aggregate(feats, code, homography, hyp, vis_c, G, normalise=True)
        ctx.save_for_backward(feats, vis_c, homography, hyp, vol_cl)
        ctx.code, ctx.G, ctx.in_dtype, ctx.channel_last = code, G, features.dtype, bool(channel_last)
        return vol_cl if channel_last else ops.cl_to_ncdhw(vol_cl)

    @staticmethod
    def backward(ctx, grad_volume):
        feats, vis, hom, hyp, vol_cl = ctx.saved_tensors
        gvol_cl = ops._f32c(grad_volume) if ctx.channel_last else ops.ncdhw_to_cl(ops._f32c(grad_volume))
        vis_sum = vis.sum(dim=1).contiguous()
        gfeat, gvis = ops.warp_corr_aggregate_bwd(feats, ctx.code, hom, hyp, vis, vis_sum, vol_cl, gvol_cl, ctx.G)
        return gfeat.to(ctx.in_dtype), gvis, None, None, None, None


# --------------------------------------------------------------------------------------------------
# native training form of the U-Net regulariser
# --------------------------------------------------------------------------------------------------
_PREC = _lib.PRECISIONS["bf16x3"]


def _block_parts(layer):
    """(conv module, BatchNorm module, is_transposed) of one Conv3d / Deconv3d wrapper or CostRegNet3D's Sequential block."""
    if isinstance(layer, nn.Sequential):
        return layer[0], layer[1], True
    return layer.conv, layer.bn, isinstance(layer.conv, nn.ConvTranspose3d)


def _stride3(conv):
    s = conv.stride
    return (s, s, s) if isinstance(s, int) else tuple(s)



def _conv_fwd(a_cl, w, stride, transposed, zero_bias, kd=3, tflip=False):
    """Linear (no bias / BatchNorm / ReLU) Conv3d(k (kd,3,3), 'same' padding, stride) or ConvTranspose3d(k3, stride (sd,2,2)) of a
    channel-last tensor on the split-bf16 MFMA kernels, from the un-folded weight tensor."""
    if transposed:
        return ops.deconv3d_linear(a_cl, ops.pack_deconv_weights_device(w, stride[0]), zero_bias, w.shape[1], stride[0], _PREC)
    if tflip:                                             # the convolution with w's taps reversed and its channel axes exchanged
        wp = ops.pack_conv_weights_device(w, packing.conv_chunk(w.shape[0], stride), tflip=True)
        return ops.conv3d_bn_relu(a_cl, wp, zero_bias, w.shape[1], kd, stride, False, _PREC)
    wp = ops.pack_conv_weights_device(w, packing.conv_chunk(w.shape[1], stride))
    return ops.conv3d_bn_relu(a_cl, wp, zero_bias, w.shape[0], kd, stride, False, _PREC)


def _conv_bwd(a_in, dz, w, stride, transposed, zero_bias, need_da=True):
    """(weight gradient, data gradient) of _conv_fwd.  Data gradients reuse the forward kernels: a transposed convolution's is the
    strided convolution with the same taps, a stride-1 convolution's the convolution with flipped, transposed taps, a strided
    convolution's the transposed convolution with the same taps."""
    if transposed:
        dw = ops.conv3d_wgrad(dz, a_in, stride)
        da = _conv_fwd(dz, w, stride, False, zero_bias) if need_da else None
        return dw, da
    dw = ops.conv3d_wgrad(a_in, dz, stride)
    if not need_da:
        return dw, None
    if stride == (1, 1, 1):
        return dw, _conv_fwd(dz, w, stride, False, zero_bias, tflip=True)
    if any(d % s for d, s in zip(a_in.shape[1:4], stride)):
        raise _lib.MvsHipError("training: strided convolutions need even input sizes (got %s)" % (tuple(a_in.shape[1:4]),))
    return dw, _conv_fwd(dz, w, stride, True, zero_bias)


def _embed_2d(w2d, cin_pad=None):
    """Conv2d weight [Cout, Cin, 3, 3] -> Conv3d weight [Cout, Cin (padded), 3, 3, 3] that acts on D = 1 volumes (only the centre
    depth tap is non-zero; the outer ones meet zero padding anyway)."""
    co, ci = w2d.shape[:2]
    w3 = torch.zeros(co, cin_pad or ci, 3, 3, 3, dtype=torch.float32, device=w2d.device)
    w3[:, :ci, 1] = w2d
    return w3


def _fast_bn(bn) -> bool:
    """The one-call block kernels cover batch-statistics BatchNorm with fp32 running statistics and a fixed momentum on one rank;
    SyncBatchNorm with an initialised process group, eval-mode BatchNorm and cumulative averaging take the granular path."""
    if not bn.training or bn.running_mean is None or bn.momentum is None:
        return False
    if isinstance(bn, nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized():
        return False
    return bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous() and bn.running_var.is_contiguous()


class _BnState:
    """Forward statistics of one BatchNorm layer (batch statistics in train mode, running statistics otherwise) and the
    bookkeeping nn.BatchNorm3d / nn.SyncBatchNorm do: momentum update of the running statistics, all-reduce of the sums."""

    def __init__(self, bn, z_cl, groups: int = 1):
        """groups > 1: `groups` equal slices of z_cl along its leading axis are normalised independently (one nn.BatchNorm call each,
        in order, in the reference): statistics come back as [groups, C]."""
        self.groups = groups
        self.batch = bn.training or bn.running_mean is None
        self.sync = False                                  # SyncBatchNorm: the sums are all-reduced over bn.process_group (None = WORLD)
        self.group = None
        C = z_cl.shape[-1]
        n_local = z_cl.numel() // C // groups
        self.count = float(n_local)
        if self.batch:
            sums = ops.bn_stats(z_cl, groups)
            if isinstance(bn, nn.SyncBatchNorm) and torch.distributed.is_available() and torch.distributed.is_initialized():
                self.sync, self.group = True, bn.process_group
                cnt = torch.tensor([float(n_local)], dtype=torch.float64, device=z_cl.device)
                torch.distributed.all_reduce(sums, group=self.group)
                torch.distributed.all_reduce(cnt, group=self.group)
                self.count = float(cnt.item())
            self.bn = bn if (bn.training and bn.running_mean is not None) else None
            if self.bn is not None and bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous() and (groups == 1 or bn.momentum is not None):
                with torch.no_grad():
                    bn.num_batches_tracked += groups
                self.mean, self.var, self.invstd = ops.bn_finalize(sums, self.count, bn.eps, bn.running_mean, bn.running_var, self._momentum())
            else:
                self.mean, self.var, self.invstd = ops.bn_finalize(sums, self.count, bn.eps)
                self.update_running_stats()
        else:
            self.bn = None
            self.mean = bn.running_mean.detach().float().contiguous()
            self.invstd = torch.rsqrt(bn.running_var.detach().float() + bn.eps).contiguous()
            if groups > 1:
                self.mean, self.invstd = self.mean.expand(groups, -1).contiguous(), self.invstd.expand(groups, -1).contiguous()

    def update_running_stats(self):
        """One momentum step of nn.BatchNorm's running statistics (unbiased variance).  Called once in the forward and once more
        in the backward: the reference runs its regulariser under torch.utils.checkpoint (module.py:393-396), whose recomputation
        in the backward pass is a second train-mode forward - its running statistics take two steps per iteration."""
        bn = self.bn
        if bn is None:
            return
        with torch.no_grad():
            if bn.running_mean.dtype == torch.float32 and bn.running_mean.is_contiguous() and (self.groups == 1 or bn.momentum is not None):
                bn.num_batches_tracked += self.groups
                ops.bn_running_update(self.mean, self.var, self.count, self._momentum(), bn.running_mean, bn.running_var)
                return
            means = self.mean.reshape(self.groups, -1)
            vars_ = self.var.reshape(self.groups, -1)
            for gi in range(self.groups):                    # one momentum step per group, in order (cumulative average: m = 1 / n)
                bn.num_batches_tracked += 1
                m = self._momentum()
                bn.running_mean.mul_(1.0 - m).add_(means[gi].to(bn.running_mean.dtype), alpha=m)
                bn.running_var.mul_(1.0 - m).add_((vars_[gi] * (self.count / max(self.count - 1.0, 1.0))).to(bn.running_var.dtype), alpha=m)

    def _momentum(self) -> float:
        bn = self.bn
        return float(bn.momentum) if bn.momentum is not None else 1.0 / float(bn.num_batches_tracked)


class RegNetTrain(torch.autograd.Function):
    """features_cl [B,D,H,W,8] = U-Net(volume_cl) up to (not including) `prob`, module.py:398-406 / 494-501, with gradients for the
    volume and all 27 parameters (nine weights, nine BatchNorm weights, nine BatchNorm biases) - see the module docstring."""

    NAMES = ("conv1", "conv2", "conv3", "conv4", "conv5", "conv6", "conv7", "conv9", "conv11")
    SKIP = {6: 3, 7: 1, 8: -1}                       # block index -> index of the block whose output is added (-1: the input volume)

    # Activations: kept by default while they are small, RECOMPUTED in the backward pass above `RECOMPUTE_ABOVE_BYTES` - the
    # reference always runs its regularisers under torch.utils.checkpoint (module.py:393-396, 488-492: only the input volume survives
    # the forward, the backward re-runs forward_once).  `reg.recompute_in_backward = True / False` overrides the size rule.  In
    # recompute mode the forward saves the input volume, the parameters and each BatchNorm's batch statistics ([C] vectors); the
    # backward re-runs convolution + normalise with THOSE statistics (the kernels are deterministic: the regenerated activations are
    # bit-identical, so are the gradients) - one extra U-Net forward without the statistics passes.
    RECOMPUTE_ABOVE_BYTES = 512 << 20

    @staticmethod
    def kept_bytes(reg, shape) -> int:
        """Bytes of the activations the keep-everything mode holds between forward and backward: every block's pre-BatchNorm output z
        and every block's output y except the last (which is the result)."""
        B, D, H, W = shape[:4]
        dims, total = {0: (D, H, W)}, 0
        for i, name in enumerate(RegNetTrain.NAMES):
            conv, _, transposed = _block_parts(getattr(reg, name))
            d, h, w = dims[i]
            dims[i + 1] = ops._out_dims(transposed, 3, _stride3(conv), d, h, w)
            od, oh, ow = dims[i + 1]
            total += (2 if i + 1 < len(RegNetTrain.NAMES) else 1) * 4 * B * od * oh * ow * conv.out_channels
        return total

    @staticmethod
    def _wants_recompute(reg, shape) -> bool:
        flag = getattr(reg, "recompute_in_backward", None)
        return bool(flag) if flag is not None else RegNetTrain.kept_bytes(reg, shape) > RegNetTrain.RECOMPUTE_ABOVE_BYTES

    @staticmethod
    def forward(ctx, volume_cl, reg, *params):
        x = ops._f32c(volume_cl.detach())
        wsp = ops.TrainWorkspace(x.device)
        zero_bias = wsp.zero_bias
        recompute = RegNetTrain._wants_recompute(reg, x.shape)
        skip_sources = {k + 1 for k in RegNetTrain.SKIP.values()}              # activation indices a later block adds
        last_use = {k + 1: i for i, k in RegNetTrain.SKIP.items()}
        blocks, acts, saved = [], {0: x}, []
        n = len(RegNetTrain.NAMES)
        for i, name in enumerate(RegNetTrain.NAMES):
            conv, bn, transposed = _block_parts(getattr(reg, name))
            w = params[3 * i].detach().float()
            gamma, beta = params[3 * i + 1].detach().float().contiguous(), params[3 * i + 2].detach().float().contiguous()
            stride = _stride3(conv)
            a_in = acts[i]
            skip_idx = RegNetTrain.SKIP.get(i)
            skip = None if skip_idx is None else acts[skip_idx + 1]
            if _fast_bn(bn):
                with torch.no_grad():
                    bn.num_batches_tracked += 1
                z, stats, y = ops.train_block_fwd(wsp, a_in, w, transposed, 3, stride, gamma, beta, bn.eps, bn.running_mean, bn.running_var,
                                                  bn.momentum, skip)
                acts[i + 1] = y
                blocks.append((transposed, stride, True, 0.0, (False, None), bn))
                saved += ([stats, stats, w, gamma, beta] if recompute else [a_in, z, stats, stats, w, gamma, beta])
            else:
                z = _conv_fwd(a_in, w, stride, transposed, zero_bias)
                st = _BnState(bn, z)
                acts[i + 1] = ops.bn_relu_apply(z, st.mean, st.invstd, gamma, beta, skip, relu=True)
                blocks.append((transposed, stride, st.batch, st.count, (st.sync, st.group), st))
                saved += ([st.mean, st.invstd, w, gamma, beta] if recompute else [a_in, z, st.mean, st.invstd, w, gamma, beta])
            if recompute:                                # drop what no later block reads: the forward's live set stays a few tensors
                del z, a_in
                for k in [k for k in acts if k <= i and k != 0 and not (k in skip_sources and last_use[k] > i)]:
                    del acts[k]
        ctx.blocks, ctx.recompute = blocks, recompute
        ctx.save_for_backward(*(([x] if recompute else []) + saved))
        return acts[n]

    @staticmethod
    def _regenerate(S, blocks, zero_bias):
        """Recompute mode: the forward again from the saved input volume with the saved batch statistics -> the (a_in, z, ...) rows the
        backward reads.  Statistics rows: one-call blocks saved [3, 1, C] (mean, var, invstd), granular blocks mean and invstd."""
        acts, rows = {0: S[0]}, []
        for i, (transposed, stride, _, _, _, bn_state) in enumerate(blocks):
            s0, s1, w, gamma, beta = S[1 + 5 * i: 6 + 5 * i]
            mean, invstd = (s0, s1) if isinstance(bn_state, _BnState) else (s0[0].contiguous(), s0[2].contiguous())
            skip_idx = RegNetTrain.SKIP.get(i)
            skip = None if skip_idx is None else acts[skip_idx + 1]
            z = _conv_fwd(acts[i], w, stride, transposed, zero_bias)
            acts[i + 1] = ops.bn_relu_apply(z, mean, invstd, gamma, beta, skip, relu=True)
            rows += [acts[i], z, s0, s1, w, gamma, beta]
        return rows

    @staticmethod
    def backward(ctx, grad_out):
        S = ctx.saved_tensors
        n = len(ctx.blocks)
        grads = [None] * (3 * n)
        pending = {}                                     # activation index (0 = volume, i + 1 = output of block i) -> gradient so far
        pending[n] = ops._f32c(grad_out)
        wsp = ops.TrainWorkspace(grad_out.device)
        zero_bias = wsp.zero_bias
        if ctx.recompute:
            S = RegNetTrain._regenerate(S, ctx.blocks, zero_bias)
        for i in range(n - 1, -1, -1):
            transposed, stride, batch, count, group, bn_state = ctx.blocks[i]
            a_in, z, mean, invstd, w, gamma, beta = S[7 * i:7 * i + 7]
            g = pending.pop(i + 1)
            skip_idx = RegNetTrain.SKIP.get(i)
            if skip_idx is not None:                     # the skip source receives the block output's gradient unchanged
                k = skip_idx + 1
                pending[k] = g if k not in pending else pending[k] + g
            if not isinstance(bn_state, _BnState):       # one-call block: bn_state is the BatchNorm module, `mean` the [3, 1, C] statistics
                bn = bn_state
                with torch.no_grad():
                    bn.num_batches_tracked += 1          # the reference's checkpoint recomputation: a second momentum step (in the call)
                grads[3 * i], grads[3 * i + 1], grads[3 * i + 2], da = ops.train_block_bwd(
                    wsp, g, a_in, z, mean, w, transposed, 3, stride, gamma, beta, bn.running_mean, bn.running_var, bn.momentum)
                pending[i] = da if i not in pending else pending[i] + da
            else:
                bn_state.update_running_stats()          # the reference's checkpoint recomputation (see _BnState)
                sums = ops.bn_relu_bwd_reduce(g, z, mean, invstd, gamma, beta, relu=True)
                grads[3 * i + 2] = sums[: sums.numel() // 2].float()                 # d beta (this rank's voxels; DDP averages)
                grads[3 * i + 1] = sums[sums.numel() // 2:].float()                  # d gamma
                if group[0]:
                    sums = sums.clone()
                    torch.distributed.all_reduce(sums, group=group[1])
                dz = ops.bn_relu_bwd_apply(g, z, mean, invstd, gamma, beta, sums, count, relu=True, use_batch_stats=batch)
                grads[3 * i], da = _conv_bwd(a_in, dz, w, stride, transposed, zero_bias)
                pending[i] = da if i not in pending else pending[i] + da
            if ctx.recompute:
                S[7 * i] = S[7 * i + 1] = None           # this block's regenerated activations are done with
        return (pending[0], None) + tuple(grads)


def regnet_forward_native(reg, volume_cl: torch.Tensor) -> torch.Tensor:
    """CostRegNet / CostRegNet3D up to `prob` on the library's kernels, channel-last in and out."""
    if not isinstance(reg.inner, nn.Identity):
        raise NotImplementedError("in_channels != base_channels (1x1x1 `inner` conv): inference only (shape-generic kernel); no training kernels")
    params = []
    for name in RegNetTrain.NAMES:
        conv, bn, _ = _block_parts(getattr(reg, name))
        if bn is None or conv.bias is not None:
            raise NotImplementedError("the training path implements the bias-free conv + BatchNorm + ReLU blocks of the regularisers")
        params += [conv.weight, bn.weight, bn.bias]
    return RegNetTrain.apply(volume_cl, reg, *params)


class VisTrain(torch.autograd.Function):
    """The three Conv2d + BatchNorm2d + ReLU blocks of the visibility CNN (cost_volume.py:36, module.py:168-197) on the library's
    kernels: entropy [B, V-1, H, W] (no gradient: it comes from sim.detach()) -> features [V-1, B, H, W, 8] channel-last, with
    gradients for the nine parameters.  The 2-D layers run as k = (1|3,3,3) convolutions on D = 1 volumes; the reference calls the
    CNN once per source view on a batch of B maps, so the BatchNorm statistics (and running-stat updates) are per view here too."""

    @staticmethod
    def forward(ctx, entropy, vis_seq, *params):
        B, NV, H, W = entropy.shape
        dev = entropy.device
        zero_bias = torch.zeros(64, dtype=torch.float32, device=dev)
        x = torch.zeros(NV, B, H, W, 8, dtype=torch.float32, device=dev)         # view-major, 8 channels: the MFMA kernels' smallest Cin
        x[..., 0] = entropy.detach().float().permute(1, 0, 2, 3)
        a = x.view(NV * B, 1, H, W, 8)
        saved, blocks = [], []
        for i in range(3):
            w2d = params[3 * i].detach().float()
            gamma, beta = params[3 * i + 1].detach().float().contiguous(), params[3 * i + 2].detach().float().contiguous()
            w3 = _embed_2d(w2d, 8 if i == 0 else None)
            if w3.shape[0] == 8:                                                 # the 16 -> 8 layer: the kernels have its k = (1,3,3) form
                z = _conv_fwd(a, w3[:, :, 1:2].contiguous(), (1, 1, 1), False, zero_bias, kd=1)
            else:
                z = _conv_fwd(a, w3, (1, 1, 1), False, zero_bias)
            st = _BnState(vis_seq[i].bn, z, groups=NV)                            # one statistics set per source view
            mean, invstd = (st.mean, st.invstd) if NV > 1 else (st.mean.reshape(1, -1), st.invstd.reshape(1, -1))
            y = ops.bn_relu_apply(z, mean, invstd, gamma, beta, None, relu=True)
            blocks.append((st.batch, st.count, (st.sync, st.group)))
            saved += [mean, invstd]
            saved += [a, z, w3, gamma, beta]
            a = y
        ctx.blocks, ctx.B, ctx.NV = blocks, B, NV
        ctx.save_for_backward(*saved)
        return a.view(NV, B, H, W, 8)

    @staticmethod
    def backward(ctx, grad_out):
        S = ctx.saved_tensors
        B, NV = ctx.B, ctx.NV
        g = ops._f32c(grad_out).view(NV * B, 1, *grad_out.shape[2:])
        zero_bias = torch.zeros(64, dtype=torch.float32, device=g.device)
        grads = [None] * 9
        for i in (2, 1, 0):
            mean, invstd, a_in, z, w3, gamma, beta = S[7 * i:7 * i + 7]
            C = z.shape[-1]
            batch, count, group = ctx.blocks[i]
            sums = ops.bn_relu_bwd_reduce(g, z, mean, invstd, gamma, beta, relu=True)             # [NV, 2C]
            grads[3 * i + 2], grads[3 * i + 1] = sums[:, :C].sum(0).float(), sums[:, C:].sum(0).float()
            if group[0]:
                sums = sums.clone()
                torch.distributed.all_reduce(sums, group=group[1])
            dz = ops.bn_relu_bwd_apply(g, z, mean, invstd, gamma, beta, sums, count, relu=True, use_batch_stats=batch)
            # weight gradient on the k = (1,3,3) form of the kernel (the maps are D = 1 volumes); data gradient = the convolution with
            # flipped, transposed taps (nothing to propagate below the first layer: the entropy carries no gradient)
            dw2 = ops.conv3d_wgrad(a_in, dz, (1, 1, 1), kd=1)
            cin = 1 if i == 0 else w3.shape[1]
            grads[3 * i] = dw2[:, :cin, 0].contiguous()
            g = _conv_fwd(dz, w3, (1, 1, 1), False, zero_bias, tflip=True) if i > 0 else None
        return (None, None) + tuple(grads)


class Prob3Train(torch.autograd.Function):
    """CostRegNet's `prob` = Conv3d(8, 1, 3, padding 1, bias=False) (module.py:391,407): features_cl [B,D,H,W,8] -> logits [B,D,H,W].
    Forward on the MFMA head kernel; the gradients reuse the 8 <-> 16 channel kernels with the single logit channel zero-padded."""

    @staticmethod
    def forward(ctx, feat_cl, weight):
        f = ops._f32c(feat_cl.detach())
        w = weight.detach().float()
        w16 = torch.zeros(16, 8, 3, 3, 3, dtype=torch.float32, device=f.device)
        w16[0] = w[0]
        zero_bias = torch.zeros(64, dtype=torch.float32, device=f.device)
        ctx.save_for_backward(f, w)
        return ops.conv3d_logits(f, ops.pack_conv_weights_device(w16, 8), zero_bias, _PREC)

    @staticmethod
    def backward(ctx, grad_logits):
        f, w = ctx.saved_tensors
        g = ops._f32c(grad_logits)
        zero_bias = torch.zeros(64, dtype=torch.float32, device=g.device)
        g8 = torch.zeros(*g.shape, 8, dtype=torch.float32, device=g.device)
        g8[..., 0] = g
        dw = ops.conv3d_wgrad(f, g8, (1, 1, 1))[0:1].contiguous()                 # [1, 8, 3, 3, 3]
        wt = torch.zeros(16, 8, 3, 3, 3, dtype=torch.float32, device=g.device)   # rows = feature channels, column 0 = the logit channel
        wt[:8, 0] = w[0].flip(1, 2, 3)
        df = _conv_fwd(g8, wt, (1, 1, 1), False, zero_bias)[..., :8].contiguous()
        return df, dw


def vis_forward_native(vis_seq, entropy: torch.Tensor) -> torch.Tensor:
    """self.vis per source view (cost_volume.py:93) -> [B, V-1, H, W]; conv / BatchNorm / ReLU blocks native, the 1x1 conv + sigmoid
    as elementwise autograd ops."""
    params = []
    for i in range(3):
        params += [vis_seq[i].conv.weight, vis_seq[i].bn.weight, vis_seq[i].bn.bias]
    t = VisTrain.apply(entropy, vis_seq, *params)                                 # [V-1, B, H, W, 8]
    last = vis_seq[3]
    v = (t * last.weight.reshape(1, 1, 1, 1, -1)).sum(-1)
    if last.bias is not None:
        v = v + last.bias
    return torch.sigmoid(v).permute(1, 0, 2, 3)


def _position_encoding_3d(position3d: torch.Tensor, C: int, rescale: float = 4.0) -> torch.Tensor:
    """PositionEncoding3D (position_encoding.py:166-189): per axis C channels, sin on the even and cos on the odd ones of
    position * rescale * 10000^(-2i/C) -> [B, 3C, D, H, W].  No gradient: the positions come from the hypotheses."""
    import math
    B, _, D, H, W = position3d.shape
    freq = torch.exp(torch.arange(0, C, 2, device=position3d.device, dtype=torch.float32) * (-math.log(10000.0) / C)).view(1, 1, -1, 1)
    ang = position3d.reshape(B, 3, 1, D * H * W).float() * rescale * freq                           # [B, 3, C/2, N]
    pe = torch.stack([torch.sin(ang), torch.cos(ang)], dim=3)                                      # [B, 3, C/2, 2, N]: interleaved
    return pe.reshape(B, 3 * C, D, H, W)


def _layer_norm_channels(x: torch.Tensor, ln) -> torch.Tensor:
    """LayerNorm3D (module.py:586-599): normalise an NCDHW tensor over its channel axis."""
    mu = x.mean(1, keepdim=True)
    xc = x - mu
    return ln.weight.view(1, -1, 1, 1, 1) * (xc * torch.rsqrt(xc.pow(2).mean(1, keepdim=True) + ln.eps)) + ln.bias.view(1, -1, 1, 1, 1)


class AttentionTrain(torch.autograd.Function):
    """a [B,n,64] = proj-less attention of one block: softmax(scale q k^T) v over all tokens, q | k | v = t W_qkv^T (no bias,
    attention.py:76-101, 141-170).  Forward = the inference kernels (mvs_tr_qkv_fwd + mvs_tr_attention_fwd: split-bf16 operands,
    flash attention on the MFMA k axis); backward = the library's attention backward (mvs_tr_attention_bwd: fp32, two launches, no
    atomics) between two plain GEMMs - q | k | v are re-projected from the saved tokens (nothing [n, n]-sized or per-key is kept
    between forward and backward), d_t = d_qkv W_qkv, d_W = d_qkv^T t."""

    @staticmethod
    def forward(ctx, t, w_qkv, heads, scale):
        tt = ops._f32c(t.detach())
        w = w_qkv.detach().float()
        a = ops.tr_attention(tt, packing.pack_linear_bf16x3(w), heads, scale, _lib.PREC_BF16X3)   # packed on the weight's device (torch ops)
        ctx.save_for_backward(tt, w, a)
        ctx.heads, ctx.scale = heads, scale
        return a

    @staticmethod
    def backward(ctx, d_a):
        t, w, a = ctx.saved_tensors
        B, n, C = t.shape
        qkv = (t.reshape(B * n, C) @ w.t()).reshape(B, n, 3 * C)                   # [.., (q | k | v), heads, 16]
        d_qkv = ops.tr_attention_bwd(qkv, a, ops._f32c(d_a), ctx.heads, ctx.scale)
        d2 = d_qkv.reshape(B * n, 3 * C)
        return (d2 @ w).reshape(B, n, C), d2.t() @ t.reshape(B * n, C), None, None


def _layer_norm_bwd(d_y, u, w, eps):
    """Backward of y = LayerNorm(u) over the last axis (weight w): -> (d_u, d_w, d_b)."""
    mu = u.mean(-1, keepdim=True)
    xc = u - mu
    rstd = torch.rsqrt(xc.pow(2).mean(-1, keepdim=True) + eps)
    xh = xc * rstd
    dxh = d_y * w
    d_u = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
    red = tuple(range(d_y.dim() - 1))
    return d_u, (d_y * xh).sum(red), d_y.sum(red)


class TransformerBlockTrain(torch.autograd.Function):
    """One post-norm block (module.py:535-583: t1 = LN1(t + g1 proj(attn(t))), t2 = LN2(t1 + g2 W2 gelu(W1 t1))) with a hand-written
    backward.  Forward = the five launches of the inference path (qkv + flash attention, proj + residual + LayerNorm, linear1 + GELU,
    linear2 + residual + LayerNorm on the split-bf16 kernels); kept for the backward: the block input, the attention output and t1 -
    the 4x wider FFN hidden layer and both pre-LayerNorm sums are recomputed.  Backward: the attention core on
    ``mvs_tr_attention_bwd``; the linears' data / weight gradients are plain GEMMs (rocBLAS through torch.mm), LayerNorm / GELU /
    residual-scale gradients element-wise tensor expressions - no autograd graph, no [n, n] tensor."""

    @staticmethod
    def forward(ctx, t, heads, scale, w_qkv, w_proj, b_proj, g1, n1w, n1b, eps1, w1, b1, w2, b2, g2, n2w, n2b, eps2):
        f = lambda p: p.detach().float().contiguous()
        pk = lambda w: packing.pack_linear_bf16x3(f(w))
        prec = _lib.PREC_BF16X3
        t0 = ops._f32c(t.detach())
        a = ops.tr_attention(t0, pk(w_qkv), heads, scale, prec)
        t1 = ops.tr_linear(a, pk(w_proj), f(b_proj), _lib.TR_EPI_RES_LN, 64, prec, residual=t0, gamma=f(g1).reshape(1), ln_w=f(n1w), ln_b=f(n1b), ln_eps=eps1)
        hdn = ops.tr_linear(t1, pk(w1), f(b1), _lib.TR_EPI_GELU, w1.shape[0], prec)
        t2 = ops.tr_linear(hdn, pk(w2), f(b2), _lib.TR_EPI_RES_LN, 64, prec, residual=t1, gamma=f(g2).reshape(1), ln_w=f(n2w), ln_b=f(n2b), ln_eps=eps2)
        ctx.save_for_backward(t0, a, t1, f(w_qkv), f(w_proj), f(b_proj), f(g1), f(n1w), f(w1), f(b1), f(w2), f(b2), f(g2), f(n2w))
        ctx.heads, ctx.scale, ctx.eps = heads, scale, (eps1, eps2)
        return t2

    @staticmethod
    def backward(ctx, d_t2):
        t0, a, t1, w_qkv, w_proj, b_proj, g1, n1w, w1, b1, w2, b2, g2, n2w = ctx.saved_tensors
        B, n, C = t0.shape
        flat = lambda x: x.reshape(B * n, -1)
        d_t2 = ops._f32c(d_t2)
        # ---- FFN half: recompute pre-GELU, hidden, the scaled branch and the pre-LayerNorm sum
        pre = flat(t1) @ w1.t() + b1
        hdn = F.gelu(pre)
        y2 = hdn @ w2.t() + b2
        d_u2, d_n2w, d_n2b = _layer_norm_bwd(flat(d_t2), flat(t1) + g2 * y2, n2w, ctx.eps[1])
        d_g2 = (d_u2 * y2).sum()
        d_y2 = g2 * d_u2
        d_w2, d_b2 = d_y2.t() @ hdn, d_y2.sum(0)
        d_pre = (d_y2 @ w2) * (0.5 * (1.0 + torch.erf(pre * 0.7071067811865476)) + pre * torch.exp(-0.5 * pre * pre) * 0.3989422804014327)
        d_w1, d_b1 = d_pre.t() @ flat(t1), d_pre.sum(0)
        d_t1 = d_u2 + d_pre @ w1
        # ---- attention half
        yp = flat(a) @ w_proj.t() + b_proj
        d_u1, d_n1w, d_n1b = _layer_norm_bwd(d_t1, flat(t0) + g1 * yp, n1w, ctx.eps[0])
        d_g1 = (d_u1 * yp).sum()
        d_yp = g1 * d_u1
        d_wp, d_bp = d_yp.t() @ flat(a), d_yp.sum(0)
        d_a = (d_yp @ w_proj).reshape(B, n, C)
        qkv = (flat(t0) @ w_qkv.t()).reshape(B, n, 3 * C)
        d_qkv = flat(ops.tr_attention_bwd(qkv, a, d_a, ctx.heads, ctx.scale))
        d_t0 = (d_u1 + d_qkv @ w_qkv).reshape(B, n, C)
        d_wqkv = d_qkv.t() @ flat(t0)
        return (d_t0, None, None, d_wqkv, d_wp, d_bp, d_g1.reshape(g1.shape), d_n1w, d_n1b, None, d_w1, d_b1, d_w2, d_b2, d_g2.reshape(g2.shape),
                d_n2w, d_n2b, None)


def transformer_forward_torch(reg, x: torch.Tensor, position3d) -> torch.Tensor:
    """PureTransformerCostReg.forward (module.py:629-646, blocks :569-581, attention dino/layers/attention.py:76-101,141-170), the
    training form of the shipped stage 1: every transformer block runs forward on the inference path's kernels and backward through
    the hand-written ``TransformerBlockTrain`` (attention backward on the library's kernel, linears as plain GEMMs, LayerNorm / GELU
    gradients as tensor expressions); the two ends - position-encoding projection + patch embedding + LayerNorm3D, and patch
    expansion + LayerNorm3D + `prob` - are PyTorch-ROCm autograd ops on the module's own parameters."""
    import math
    if reg.training and (reg.drop or reg.attn_drop):
        raise NotImplementedError("dropout inside the transformer regulariser (drop / attn_drop != 0) is not used by the shipped config")
    if position3d is not None:
        x = x + reg.pe_proj(_position_encoding_3d(position3d, x.shape[1]))
    x = _layer_norm_channels(reg.down[0](x), reg.down[1])
    B, C, d, h, w = x.shape
    heads = reg.num_heads
    t = x.permute(0, 3, 4, 2, 1).reshape(B, h * w * d, C)                                          # tokens ordered (h w d), module.py:573
    N = t.shape[1]
    scale = (C // heads) ** -0.5
    if reg.softmax_scale == "entropy_invariance":
        scale *= math.log(N, reg.train_avg_length)
    for blk in reg.attention_layers:
        if C == 64 and heads == 4 and blk.attn.proj.bias is not None and blk.ffn.linear1.bias is not None and blk.ffn.linear2.bias is not None:
            t = TransformerBlockTrain.apply(t, heads, scale, blk.attn.qkv.weight, blk.attn.proj.weight, blk.attn.proj.bias, blk.gamma1,
                                            blk.norm1.weight, blk.norm1.bias, blk.norm1.eps, blk.ffn.linear1.weight, blk.ffn.linear1.bias,
                                            blk.ffn.linear2.weight, blk.ffn.linear2.bias, blk.gamma2, blk.norm2.weight, blk.norm2.bias, blk.norm2.eps)
            continue
        a = AttentionTrain.apply(t, blk.attn.qkv.weight, heads, scale)               # bias-free variants: attention core native, the rest autograd ops
        proj, l1, l2 = blk.attn.proj, blk.ffn.linear1, blk.ffn.linear2
        t = blk.norm1(t + blk.gamma1 * proj(a))
        t = blk.norm2(t + blk.gamma2 * l2(F.gelu(l1(t))))
    x = t.reshape(B, h, w, d, C).permute(0, 4, 3, 1, 2)
    return reg.prob(_layer_norm_channels(reg.up[0](x), reg.up[1]))


_FP32_TRAIN_WARNED = False


def stage_forward_train(net, features, proj_matrices, depth_values, tmp, position3d=None) -> Dict[str, torch.Tensor]:
    """StageNet.forward with autograd (cost_volume.py:51-133).  See the module docstring for what runs where."""
    from .module import PureTransformerCostReg
    transformer = isinstance(net.cost_reg, PureTransformerCostReg)
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
    if G != 8:
        # inference runs such a stage on the shape-generic convolution kernel (module._RegNetBase.forward_cl_generic); the training
        # kernels (BatchNorm statistics, weight gradients, data gradients) exist for the tuned tables' widths only
        raise NotImplementedError("base_ch=%d: the HIP training kernels are built for 8 groups (all shipped configs); inference of this "
                                  "stage works (shape-generic kernel), fine-tuning it does not" % G)
    prec = getattr(net, "conv_precision", "bf16x3")
    if prec == "fp32":
        # a head configured for exact-fp32 INFERENCE can still be fine-tuned: the training kernels contract in split bf16
        # (three terms: 1e-6 from fp32) with fp32-MFMA weight gradients and fp32 activations - said once, not refused
        global _FP32_TRAIN_WARNED
        if not _FP32_TRAIN_WARNED:
            import warnings
            _FP32_TRAIN_WARNED = True
            warnings.warn("mvsformerplusplus_amd: conv_precision='fp32' selects the exact contraction at inference only; the training path runs "
                          "the fp32-equivalent split-bf16 kernels (forward / data gradients) and fp32-MFMA weight gradients.", RuntimeWarning, stacklevel=3)
    elif prec not in ("bf16x3",) + _lib.F16_FORMATS:       # the fp16 formats are inference storage formats: training keeps fp32 activations on the bf16x3 kernels
        raise NotImplementedError("conv_precision=%r: the native training kernels contract in split bf16 (forward and data gradients) "
                                  "and fp32 MFMA (weight gradients)" % net.conv_precision)
    vis = vis_forward_native(net.vis, entropy)                                                    # [B,V-1,H,W]
    if transformer:
        volume = WarpCorrAggregate.apply(features, vis, hom, hyp, G)                               # [B,G,D,H,W]
        prob_volume_pre = transformer_forward_torch(net.cost_reg, volume, position3d).squeeze(1)
    else:
        volume_cl = WarpCorrAggregate.apply(features, vis, hom, hyp, G, True)                      # [B,D,H,W,G]
        feat_cl = regnet_forward_native(net.cost_reg, volume_cl)                                   # [B,D,H,W,8]
        prob = net.cost_reg.prob
        if tuple(prob.kernel_size) == (1, 1, 1):
            prob_volume_pre = (feat_cl * prob.weight.reshape(1, 1, 1, 1, -1)).sum(-1)              # module.py:486,502
            if prob.bias is not None:
                prob_volume_pre = prob_volume_pre + prob.bias
        else:
            prob_volume_pre = Prob3Train.apply(feat_cl, prob.weight)                               # module.py:391,407
    return stage_head_train(net, prob_volume_pre, depth_values, tmp)


def stage_head_train(net, prob_volume_pre, depth_values, tmp) -> Dict[str, torch.Tensor]:
    """softmax / regression / confidence of a stage in autograd ops (cost_volume.py:105-131), from the logits [B,D,H,W]."""
    prob_volume = F.softmax(prob_volume_pre, dim=1)
    D = prob_volume_pre.shape[1]
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
