"""Mirror of the reference's ``models/cost_volume.py``: one cascade stage behind the same ``nn.Module`` API.

``StageNet(args, ndepth, stage_idx).forward(features, proj_matrices, depth_values, tmp, position3d=None)`` takes and
returns exactly what the reference does (cost_volume.py:21-133) and owns the same parameters (``vis.*``,
``cost_reg.*``), but the work is seven HIP launches on the caller's stream instead of ~150 ATen ops:

    compose_homography -> warp_corr_entropy (all views) -> vis CNN (all views) -> warp_corr_aggregate
    -> regulariser U-Net (9 MFMA conv launches) -> prob + softmax + regression + confidence

With ``view_group`` set (a torch.distributed process group over RCCL) the source views are sharded over the
ranks of the group.  Per stage the partial ``volume_sum`` / ``vis_sum`` are combined either with
ONE all-reduce (coarse stages: everything after it is replicated) or, where a row slab is at least one halo tall, by a
slab exchange after which every rank regularises 1 / R of the volume (plus a 40-row halo) and an all-gather of the per-row
outputs gives every rank the whole stage result - bit-identical on all ranks in both forms (``shard_mode``).
In train mode / with features that require grad the stage runs the native training path of ``training.py``.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib, ops, packing
from .module import (DEFAULT_PRECISION, DEFAULT_STAGE_POLICY, resolve_stage_precision, F16_FORMATS, MFMA_FORMATS, ConvBnReLU, CostRegNet, CostRegNet3D, PureTransformerCostReg, _bn_dict, _no_grad_path,
                     _PackedCache, precision_code)


def shard_views(n_src: int, world: int, rank: int):
    """Contiguous, balanced split of source views 1..n_src over `world` ranks -> [begin, end) (1-based view ids)."""
    base, extra = divmod(n_src, world)
    begin = 1 + rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


# Default precision POLICY of a stage at inference = module.DEFAULT_STAGE_POLICY ("stagemix", round 5): the coarse stages
# (ndepth > model_th) fp32-equivalent - "bf16x3" regulariser + visibility CNN, exact gather -, the CostRegNet3D stages "f16mix" (fp16 U-Net
# tensors, fp16 gather forms).  Depth vs the fp32 oracle ~5e-6 plain / 2e-5 on the x30-logits stress set / 2-4e-4 on cfg4 / cfg5's
# ill-conditioned literal range (bar 1e-3).  args["conv_precision"] overrides it per head with the policy name or ONE format for every
# stage ("f16mix", "f16x2", "f16", "bf16x3", "fp32"; module.py).  Training always runs the bf16x3 kernels on fp32 activations.
STAGE_DEFAULT_PRECISION = DEFAULT_STAGE_POLICY


_F16_CALLS = {}                            # fp16-format stage calls per device index (one process may drive several GPUs)
_F16_CHECK_DUE = set()                     # devices whose check fell inside a hipGraph capture: done at the next eager call
KEEP_CORRELATIONS_MAX_BYTES = 1 << 30      # largest kept per-view correlation tensor ([B,V-1,D,H,W,8] fp16 / fp32) of the streaming pass 2
KEEP_MIN_DEPTH = 5                         # fp16 gather form: pass 1 keeps fp16 correlations from this many planes on (D <= 4: see StageNet.keep_min_depth)
KEEP_EXACT_MIN_DEPTH = 16                  # exact gather (gather_precision "f32"): fp32 kept correlations pay from this many planes on (32 B per
                                           # voxel and view streamed twice against a second gather; D = 8: 453 MB at cfg2's stage 3 - no gain)
F16_SATURATION_CHECK_EVERY = 4096          # stage calls between two automatic reads of the saturation counter (0 = never)


def check_f16_saturation(device=None, warn: bool = True, reset: bool = True) -> int:
    """Read (and by default clear) the library's fp16 saturation counter on `device`: the number of work-items that stored a regulariser
    ACTIVATION beyond the fp16 range (clamped to +-65504) in the fp16 formats (the default "f16mix", "f16x2", "f16").  0 on sane weights; anything else means the
    default degraded values that conv_precision="bf16x3" would have kept - a warning says so once per process.  Synchronises the device:
    StageNet calls it by itself after its 8th inference call and then every F16_SATURATION_CHECK_EVERY calls (never during graph capture)."""
    import warnings
    n = ops.f16_saturation_count(reset=reset, device=device)
    if n and warn:
        warnings.warn("mvsformerplusplus_amd: %d work-items stored fp16 regulariser activations beyond +-65504 (clamped).  The default "
                      "fp16 regulariser format is degrading this model's values; build the stages with conv_precision='bf16x3' "
                      "(fp32-equivalent activations) for it." % n, RuntimeWarning, stacklevel=2)
    return n


_HYP_WARNED = False


def check_hypothesis_conditioning(hyp: torch.Tensor, warn: bool = True) -> float:
    """Fraction of depth hypotheses that are not finite or not positive.  The reference's inverse-depth schedule (module.py:712-716) takes
    1 / depth -/+ ratio * interval; on a wide range (depth_max / depth_min > (ndepths[0] - 1) / ratio + 1, e.g. 0.5 .. 10 with 32 planes) the
    window crosses zero for far pixels: their hypotheses jump through +-infinity, the result there means nothing in ANY arithmetic, and
    the pixels around them are ill-conditioned - a 6e-5 perturbation (the fp16 formats' noise) becomes 3.6e-3 mean relative depth error on
    BASELINE cfg4's literal range, where conv_precision="bf16x3" stays at 2e-4.  One warning per process says so.
    Synchronises the device; StageNet calls it with the saturation check (8th call, then every F16_SATURATION_CHECK_EVERY calls)."""
    global _HYP_WARNED
    bad = float((~torch.isfinite(hyp) | (hyp <= 0)).float().mean())
    if bad > 0 and warn and not _HYP_WARNED:
        import warnings
        _HYP_WARNED = True
        warnings.warn("mvsformerplusplus_amd: %.1f %% of the depth hypotheses of a stage are non-finite or non-positive - the inverse-depth "
                      "schedule crossed zero (depth range too wide for the number of planes).  Depth there is meaningless and ill-conditioned "
                      "around it; a uniform fp16 regulariser format loses accuracy on such inputs (build the stages with conv_precision='bf16x3', or leave "
                      "conv_precision at its default policy 'stagemix')."
                      % (100.0 * bad), RuntimeWarning, stacklevel=2)
    return bad


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
        # "auto": all-reduce of the partial volumes on coarse stages, H-slab exchange + 1/R of the regulariser where a slab is
        # at least one halo tall; "allreduce" / "slab" force one form
        self.shard_mode = "auto"
        self.keep_correlations = "auto"   # pass 1 keeps per-view correlations, pass 2 streams them (_keeps_correlations)
        self.keep_min_depth = None        # None = KEEP_MIN_DEPTH / KEEP_EXACT_MIN_DEPTH; fewer planes gather twice
        self.fuse_prob_head = True        # CostRegNet3D + bf16x3: `prob` applied in the last deconvolution's epilogue
        self.last_collective_bytes = 0
        self._buffers_cache = {}
        self.return_prob_volumes = True   # prob_volume / prob_volume_pre are only read by the training losses
        # args["conv_precision"]: the policy "stagemix" (the default, module.DEFAULT_STAGE_POLICY) or one format for every stage; resolved
        # per stage here - conv_precision is always a concrete format ("bf16x3" | "f16mix" | ...), gather_precision "f16" (fp16 windows +
        # fp16 kept correlations) or "f32" (exact: fp32 windows, fp32 kept correlations or a second gather)
        self.conv_precision = args.get("conv_precision", STAGE_DEFAULT_PRECISION)        # property: resolves the policy for this stage
        self._vis_cache = _PackedCache()

    @property
    def conv_precision(self) -> str:
        """The stage's concrete regulariser / visibility-CNN format.  Assigning a policy ("stagemix") or a format re-resolves
        `precision_policy`, `conv_precision` and `gather_precision` for this stage (module.resolve_stage_precision)."""
        return self._conv_precision

    @conv_precision.setter
    def conv_precision(self, policy: str) -> None:
        self.precision_policy = policy
        self._conv_precision, self.gather_precision = resolve_stage_precision(policy, self.ndepth, self.args.get("model_th", 8),
                                                                              bool(self.args.get("final_stage", False)))

    # ---- packed parameters ----
    def _vis_params(self, device):
        prec = self._vis_precision()
        if prec in F16_FORMATS:
            pack = lambda w, ch: packing.f16x2(packing.pack_conv_weights_bf16x3, w, ch)
        else:
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

    def _vis_precision(self) -> str:
        """Contraction of the visibility CNN's two MFMA layers (its activations stay on chip): the stage's conv_precision - "f16x2" runs
        the fp16 two-term form with fp16 rings."""
        return self.conv_precision           # every fp16 format shares the fp16 rings; "f16" / "f16mix" drop the CNN's second weight term too

    def _keeps_correlations(self, feats, G, hyp) -> bool:
        """Pass 2 as a stream over correlations kept by pass 1 (ops.warp_corr_entropy_keep / corr_aggregate) instead of a second
        gather: `keep_correlations` True / "auto" (default) where the library builds it (LDS-staged shapes, D > 4) and the kept tensor
        fits KEEP_CORRELATIONS_MAX_BYTES; False = always gather twice.  gather_precision "f16" keeps fp16 correlations (16 B per voxel
        and view), "f32" fp32 ones (32 B: exact - from KEEP_EXACT_MIN_DEPTH planes on, where the stream is cheaper than the gather)."""
        if not self.keep_correlations:
            return False
        exact = self.gather_precision != "f16"
        B, V, _, H, W = feats.shape
        D = hyp.shape[1]
        dmin = self.keep_min_depth if self.keep_min_depth is not None else (KEEP_EXACT_MIN_DEPTH if exact else KEEP_MIN_DEPTH)
        if D < dmin or (exact and D <= 4):
            return False
        if (V - 1) * B * D * H * W * (32 if exact else 16) > KEEP_CORRELATIONS_MAX_BYTES:
            return False
        return ops.gather_keeps_correlations(feats, G, hyp)

    def _f16_activations(self) -> bool:
        """conv_precision "f16x2": the U-Net's tensors - cost volume included - are fp16 in HBM (MVS_PREC_F16X2); the transformer
        regulariser reads an fp32 volume and is unaffected."""
        return self.conv_precision in F16_FORMATS and not isinstance(self.cost_reg, PureTransformerCostReg) and not self._generic_regulariser()

    def _generic_regulariser(self) -> bool:
        """base_ch != 8: the reference builds CostRegNet(G, G) / CostRegNet3D(G, G) (cost_volume.py:44-49) - widths no tuned MFMA kernel
        exists for.  The stage then gathers with the direct kernels (any G dividing C), keeps an fp32 volume [B,D,H,W,G] and regularises
        it layer by layer on the shape-generic exact-fp32 convolution (module._RegNetBase.forward_cl_generic); conv_precision still
        selects the visibility CNN's format."""
        return bool(getattr(self.cost_reg, "is_generic", False))

    def _wants_autograd(self, features) -> bool:
        """Training mode (BatchNorm batch statistics) or a caller that differentiates w.r.t. the features."""
        return self.training or (torch.is_grad_enabled() and torch.is_tensor(features) and features.requires_grad)

    def forward(self, features, proj_matrices, depth_values, tmp, position3d=None, _fused: Optional[dict] = None) -> Dict[str, torch.Tensor]:
        """The reference's signature (cost_volume.py:51).  `_fused` is CascadeDepthHead's private side channel (round 5, fewer launches):
        "homography" = this stage's [B,V-1,12] homographies from the cascade's one-launch prologue (otherwise composed here);
        "conf_prev" = the earlier stages' confidence maps - the head then also writes the cascade's averaged confidence into
        _fused["conf_avg"] (a16 fused into the last stage's head); "next" = (ndepth, ratio) of the NEXT stage - the head then also writes
        that stage's inverse-depth hypotheses [B,ndepth,2H,2W] into _fused["next_hyp"] (a14 fused, module.py:707-724).  The returned dict
        is the reference's five keys either way."""
        if self._wants_autograd(features):
            # autograd Functions over the library's training kernels (gather forward / backward, U-Net and
            # visibility CNN convolutions, BatchNorm, weight gradients); training.py says exactly what runs where
            from .training import stage_forward_train
            assert features.shape[1] == proj_matrices.shape[1], "Different number of images and projection matrices"
            return stage_forward_train(self, features, proj_matrices, depth_values, tmp, position3d)
        B, V, C, H, W = features.shape
        assert V == proj_matrices.shape[1], "Different number of images and projection matrices"   # cost_volume.py:56
        G = self.in_channels
        if G > C:
            raise AssertionError("G must <= C!")                                                  # cost_volume.py:87
        if C % G != 0:
            raise AssertionError("base_ch=%d must divide the %d feature channels (the reference's .view(B, G, C // G, ...), cost_volume.py:80)" % (G, C))
        feats, code = ops._feat(features)
        hyp = ops._f32c(depth_values)
        if hyp.dim() == 2:
            # [B, D]: fronto-parallel planes shared by every pixel (a plain MVSNet call; the reference's warp broadcasts them, warping.py:91,
            # and depth_regression views them as [B, D, 1, 1], module.py:650-652).  The kernels read per-pixel hypotheses: expand once.
            hyp = hyp[:, :, None, None].expand(B, hyp.shape[1], H, W).contiguous()
        elif hyp.dim() != 4:
            raise ValueError("depth_values must be [B,D] or [B,D,H,W], got shape %s" % (tuple(depth_values.shape),))
        hom = _fused.get("homography") if _fused else None
        if hom is None:
            hom = ops.compose_homography(proj_matrices)
        conf_prev = _fused.get("conf_prev") if _fused and self.view_group is None else None
        nxt = _fused.get("next") if _fused and self.view_group is None and conf_prev is None and hyp.shape[1] >= 3 else None
        vis_params = self._vis_params(feats.device)
        prec = precision_code(self._vis_precision())
        if self.view_group is not None:
            import torch.distributed as dist
            world = dist.get_world_size(self.view_group)
            slab = self._slab_plan(H, world) if self.shard_mode in ("auto", "slab") else None
            if slab is not None and not isinstance(self.cost_reg, PureTransformerCostReg):
                return self._forward_slab(feats, code, hom, hyp, depth_values, G, vis_params, float(tmp), slab)
            volume = self._sharded_volume(feats, code, hom, hyp, G, vis_params)
            split = self._split_activations()
        else:
            split = self._split_activations()
            f16 = self._f16_activations()
            w16 = self.gather_precision == "f16"
            if self._keeps_correlations(feats, G, hyp):
                # pass 1 keeps the per-view group correlations (fp16 / 16 B per voxel and view in the fp16 gather form, fp32 / 32 B in the
                # exact one) and pass 2 streams them - cheaper than the second gather, which is bound by window staging and LDS reads
                # (DESIGN.md 4.1).  The volume comes out in the regulariser's own format (fp16 / split bf16 / fp32), whatever the gather keeps
                entropy, corr = ops.warp_corr_entropy_keep(feats, code, hom, hyp, G, exact=not w16)
                vis = ops.vis_weight(entropy, vis_params, prec)
                volume = ops.corr_aggregate(corr, vis, split=split, f16=f16)
                del corr
            else:
                # pass 1 -> visibility CNN -> pass 2 gathers again and writes the cost volume once (no per-view intermediate in HBM)
                entropy = ops.warp_corr_entropy(feats, code, hom, hyp, G, f16_window=w16)
                vis = ops.vis_weight(entropy, vis_params, prec)
                if f16 and not ops.gather_is_lds_staged(feats, G, hyp):   # shapes the LDS-staged gather does not cover: fp32 volume, converted
                    volume = ops.volume_to_f16(ops.warp_corr_aggregate(feats, code, hom, hyp, vis, G, normalise=True)[0])
                else:
                    volume, _ = ops.warp_corr_aggregate(feats, code, hom, hyp, vis, G, normalise=True, split=split, f16=f16)
        out = self._regularise_and_regress(volume, hyp, depth_values, float(tmp), position3d, split=split, conf_prev=conf_prev, nxt=nxt)
        if conf_prev is not None:
            _fused["conf_avg"] = out.pop("_conf_avg")
        if "_next_hyp" in out:
            _fused["next_hyp"] = out.pop("_next_hyp")
        if self._f16_activations():
            self._count_f16_call(feats.device, hyp)
        return out

    @staticmethod
    def _count_f16_call(device, hyp=None):
        """Automatic checks of the fp16 formats (check_f16_saturation, check_hypothesis_conditioning): after the 8th fp16-format inference
        call ON THIS DEVICE, then every F16_SATURATION_CHECK_EVERY calls; one device synchronisation each time.  A check that falls inside
        a hipGraph capture is deferred to the next eager call on the device, not dropped."""
        key = device.index if device.type == "cuda" else -1
        n = _F16_CALLS[key] = _F16_CALLS.get(key, 0) + 1
        due = bool(F16_SATURATION_CHECK_EVERY) and (n == 8 or n % F16_SATURATION_CHECK_EVERY == 0 or key in _F16_CHECK_DUE)
        if not due:
            return
        if device.type == "cuda" and torch.cuda.is_current_stream_capturing():
            _F16_CHECK_DUE.add(key)
            return
        _F16_CHECK_DUE.discard(key)
        check_f16_saturation(device)
        if hyp is not None:
            check_hypothesis_conditioning(hyp)

    def _split_activations(self) -> bool:
        """The bf16x3 U-Net keeps its activations - cost volume included - in the split hi | lo bf16 format between layers
        (MVS_PREC_BF16X3_SPLIT, csrc/conv_bf16x3_kernels.hip); the transformer regulariser and the fp32 contraction read fp32."""
        return self.conv_precision == "bf16x3" and not isinstance(self.cost_reg, PureTransformerCostReg) and not self._generic_regulariser()

    def _head_mode(self, D):
        conf_n = 0
        if self.depth_type == "ce":
            mode = _lib.HEAD_CE_TRAIN if self.training else _lib.HEAD_CE_EVAL
        else:
            mode = _lib.HEAD_REG
            conf_n = 4 if D >= 32 else (3 if D == 16 else (2 if D == 8 else 0))                    # cost_volume.py:121-128
        return mode, conf_n

    def _regularise_and_regress(self, volume, hyp, depth_values, tmp, position3d=None, split=False, conf_prev=None, nxt=None) -> Dict[str, torch.Tensor]:
        """cost_volume.py:103-131 on a normalised channel-last volume [B,D,H,W,8] (H may be a row slab of the stage); split: the
        volume is in the split activation format and the U-Net runs MVS_PREC_BF16X3_SPLIT.  conf_prev: the earlier stages' confidence
        maps - the head also averages them with this stage's (extra key "_conf_avg", popped by forward); nxt = (ndepth, ratio): the head
        also schedules the next stage's hypotheses (extra key "_next_hyp")."""
        D = hyp.shape[1]
        pcode = _lib.PREC_BF16X3_SPLIT if split else precision_code(self.conv_precision)
        mfma = self.conv_precision in MFMA_FORMATS
        mode, conf_n = self._head_mode(D)
        conf_avg = None

        def head(logits):
            if nxt is not None:
                return ops.softmax_regress_schedule(logits, hyp, tmp, mode, conf_n, self.return_prob_volumes, int(nxt[0]), float(nxt[1]))
            return ops.softmax_regress(logits, hyp, tmp, mode, conf_n, self.return_prob_volumes, conf_prev=conf_prev)

        if isinstance(self.cost_reg, PureTransformerCostReg):
            prob_volume_pre = self.cost_reg.logits_cl(volume, position3d)
            depth, conf, prob_volume, *rest = head(prob_volume_pre)
        elif self._generic_regulariser():
            prob_volume_pre = self.cost_reg.logits_cl_generic(volume)
            depth, conf, prob_volume, *rest = head(prob_volume_pre)
        else:
            ws, bs, prob_w, prob_b = self.cost_reg.packed_all(volume.device, self.conv_precision)
            if self.cost_reg.prob_ksize == 1 and mfma and self.fuse_prob_head:
                # CostRegNet3D: the 1x1x1 head rides in the last deconvolution's epilogue (module.py:500-502): logits out, the
                # 8-channel full-resolution features never reach HBM
                prob_volume_pre = ops.regnet_logits(self.cost_reg.kind, volume, ws, bs, prob_w, prob_b, pcode)
                depth, conf, prob_volume, *rest = head(prob_volume_pre)
            elif self.cost_reg.prob_ksize == 3 and mfma:
                # CostRegNet: the 3x3x3 head (module.py:391,407) as an MFMA convolution with one real output row, logits out
                feat_cl = ops.regnet(self.cost_reg.kind, volume, ws, bs, pcode)
                prob_volume_pre = ops.conv3d_logits(feat_cl, prob_w, prob_b, pcode)
                depth, conf, prob_volume, *rest = head(prob_volume_pre)
            else:
                feat_cl = ops.regnet(self.cost_reg.kind, volume, ws, bs, pcode)
                if split:                                       # fuse_prob_head = False (A/B switch): the standalone head reads fp32
                    feat_cl = ops.from_split(feat_cl)
                elif feat_cl.dtype == torch.float16:
                    feat_cl = feat_cl.float()
                depth, conf, prob_volume, prob_volume_pre = ops.prob_regress(
                    feat_cl, prob_w, prob_b, self.cost_reg.prob_ksize, hyp, tmp, mode, conf_n, self.return_prob_volumes)
                rest = []
                if conf_prev is not None:                       # the fp32-exact route has no fused form: the stand-alone average
                    rest = [ops.confidence_average(list(conf_prev) + [conf], conf.shape[-2], conf.shape[-1])]
        out = {"depth": depth, "prob_volume": prob_volume, "photometric_confidence": conf,
               "depth_values": depth_values, "prob_volume_pre": prob_volume_pre}
        if conf_prev is not None:
            out["_conf_avg"] = rest[0]
        elif nxt is not None and rest:
            out["_next_hyp"] = rest[0]
        return out

    # ---- source views sharded over the ranks of `view_group` ---------------------------------------
    MAX_CACHED_SHAPES = 2          # input resolutions whose scratch is kept (mixed-resolution datasets would otherwise grow it without bound)

    def clear_buffers(self):
        """Free the persistent scratch of the sharded path."""
        self._buffers_cache.clear()

    def _buffer(self, name, shape, device, zero=False):
        """Persistent scratch per (name, shape): the sharded path allocates nothing in steady state.  One buffer per name is kept
        for each of the MAX_CACHED_SHAPES most recently used shapes (LRU)."""
        # per stream: two reference views in flight through the same module (separate streams) must not share scratch
        stream = torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0
        key = (name, tuple(shape), device, stream)
        buf = self._buffers_cache.pop(key, None)
        if buf is None:
            same = [k for k in self._buffers_cache if k[0] == name and k[2] == device and k[3] == stream]
            for k in same[: max(0, len(same) - (self.MAX_CACHED_SHAPES - 1))]:        # dicts keep insertion order: oldest first
                del self._buffers_cache[k]
            buf = torch.empty(shape, dtype=torch.float32, device=device)
        self._buffers_cache[key] = buf                              # (re-)inserted last = most recently used
        if zero:
            buf.zero_()
        return buf

    def _partial_volume(self, feats, code, hom, hyp, G, vis_params, flat):
        """This rank's share of volume_sum / vis_sum (cost_volume.py:97-98) written into `flat` = [vol | vsum]."""
        import torch.distributed as dist
        B, V, C, H, W = feats.shape
        D = hyp.shape[1]
        world, rank = dist.get_world_size(self.view_group), dist.get_rank(self.view_group)
        vb, ve = shard_views(V - 1, world, rank)
        nvol = B * D * H * W * G
        vol = flat[:nvol].view(B, D, H, W, G)
        vsum = flat[nvol:].view(B, H, W)
        if ve > vb:
            # The sharded passes ALWAYS gather with fp32 source windows and sum fp32 partial volumes (no kept correlations: pass 2 must
            # produce the un-normalised per-rank partial sum), whatever gather_precision says - on the fp16-format stages a sharded head is
            # therefore slightly MORE exact than the single-GPU one (fp16 windows / fp16 kept correlations there): they agree to an fp16 ulp
            # of the volume, not bit for bit (tests/test_dist_gloo.py asserts that bound).  ADVICE r4.
            # entropy / visibility maps are indexed by absolute view; only this rank's views are written and read
            entropy = self._buffer("entropy", (B, V - 1, H, W), feats.device)
            ops.warp_corr_entropy(feats, code, hom, hyp, G, vb, ve, out=entropy)
            vis = self._buffer("vis", (B, V - 1, H, W), feats.device)
            own = entropy[:, vb - 1: ve - 1]
            vis[:, vb - 1: ve - 1] = ops.vis_weight(own if own.is_contiguous() else own.contiguous(), vis_params, precision_code(self._vis_precision()))
            ops.warp_corr_aggregate(feats, code, hom, hyp, vis, G, normalise=False, view_begin=vb, view_end=ve, out=(vol, vsum))
        else:
            flat.zero_()                                     # more ranks than source views: this rank contributes nothing
        return vol, vsum

    def _sharded_volume(self, feats, code, hom, hyp, G, vis_params):
        """All-reduce mode: ONE all-reduce(sum) of [G*D*HW + HW] floats per stage, regulariser replicated on every rank."""
        import torch.distributed as dist
        B, V, C, H, W = feats.shape
        D = hyp.shape[1]
        flat = self._buffer("partial", (B * D * H * W * G + B * H * W,), feats.device)
        vol, vsum = self._partial_volume(feats, code, hom, hyp, G, vis_params, flat)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.view_group)
        self.last_collective_bytes = flat.numel() * 4
        if self._f16_activations():
            return ops.volume_to_f16(vol, vsum)                  # normalise + convert, out of place
        return ops.volume_normalise_(vol, vsum, split=self._split_activations())

    # slab mode -----------------------------------------------------------------------------------
    SLAB_HALO = 40      # rows: >= the U-Net + prob receptive radius (1+2+2+4+4+8+8+4+2+1 = 36) and a multiple of 8 (stride phase)

    def _slab_plan(self, H, world):
        """Row slabs [a, b) of equal height S (multiple of 8) per rank and their halo-extended ranges, or None when the slabs
        would be thinner than the halo (the exchange + redundant halo work would not pay: coarse stages stay all-reduce)."""
        S = -(-H // world)
        S = -(-S // 8) * 8
        if self.shard_mode != "slab" and S < self.SLAB_HALO:
            return None
        plan = []
        for r in range(world):
            a, b = min(r * S, H), min((r + 1) * S, H)
            ea, eb = max(0, a - self.SLAB_HALO), min(H, b + self.SLAB_HALO)
            plan.append((a, b, ea, eb))
        return S, plan

    def _forward_slab(self, feats, code, hom, hyp, depth_values, G, vis_params, tmp, slab):
        """Partial volumes -> every rank receives the partials of ITS halo-extended row slab from all ranks, sums them,
        regularises 1 / world of the volume (plus halo) and regresses its rows; an all-gather of the per-row outputs gives every
        rank the whole stage result (bit-identical on all ranks: each row has one owner)."""
        import torch.distributed as dist
        B, V, C, H, W = feats.shape
        D = hyp.shape[1]
        dev = feats.device
        world, rank = dist.get_world_size(self.view_group), dist.get_rank(self.view_group)
        S, plan = slab
        flat = self._buffer("partial", (B * D * H * W * G + B * H * W,), dev)
        vol, vsum = self._partial_volume(feats, code, hom, hyp, G, vis_params, flat)
        a, b, ea, eb = plan[rank]
        rows = eb - ea
        # ---- exchange: to rank j the rows [ea_j, eb_j) of this rank's partial (volume rows + vis_sum rows in one message).  ONE launch
        #      packs the messages of all destinations (mvs_slab_pack), ONE launch sums the own slice and every received message in rank
        #      order (mvs_slab_reduce): no staging copies of the own slab, no R - 1 add_ launches ----
        p2p, sends, recvs = [], [None] * world, [None] * world
        for j in range(world):
            ja, jb, jea, jeb = plan[j]
            if jb <= ja or j == rank:
                continue                                     # rank j owns no rows (more ranks than 8-row slabs) / no message to self
            sends[j] = self._buffer("send%d" % j, (B * D * (jeb - jea) * W * G + B * (jeb - jea) * W,), dev)
        if any(b is not None for b in sends):
            ops.slab_pack(vol, vsum, sends, [(pl[2], pl[3]) for pl in plan])
        for j in range(world):
            if sends[j] is not None:
                p2p.append(dist.P2POp(dist.isend, sends[j], dist.get_global_rank(self.view_group, j), self.view_group))
        nmine = B * D * rows * W * G + B * rows * W
        if b > a:
            for j in range(world):
                if j == rank:
                    continue
                recvs[j] = self._buffer("recv%d" % j, (nmine,), dev)
                p2p.append(dist.P2POp(dist.irecv, recvs[j], dist.get_global_rank(self.view_group, j), self.view_group))
        if p2p:
            # Overlap comes from the caller keeping two or more reference views in flight per group on separate streams (bench.py issues
            # them round-robin): the scratch buffers are per stream, the collectives of a view are enqueued behind its own launches
            # only, and every rank issues the views in the same order.
            # Stream safety on RCCL (backend "nccl"): ProcessGroupNCCL runs the grouped send / recv on its own stream, ordered behind an
            # event recorded on the CURRENT stream at this call (so slab_pack's writes are seen), and work.wait() makes the current stream
            # wait for the communication - every later kernel of this view (slab_reduce, and the next call's slab_pack into the same send
            # buffers) is ordered behind it.  The buffers come from _buffer(): persistent per (name, shape, device, issuing stream), never
            # handed back to the caching allocator while a collective can touch them, never shared between streams - the cases
            # Tensor.record_stream exists for (free / reuse by another stream) cannot arise.
            for w in dist.batch_isend_irecv(p2p):
                w.wait()
        self.last_collective_bytes = sum(t.numel() for t in sends if t is not None) * 4
        out_rows = None
        if b > a:
            acc = ops.slab_reduce(vol, vsum, recvs, rank, self._buffer("slab", (nmine,), dev), ea, eb)
            svol = acc[: nmine - B * rows * W].view(B, D, rows, W, G)
            ssum = acc[nmine - B * rows * W:].view(B, rows, W)
            if self._f16_activations():
                svol = ops.volume_to_f16(svol, ssum)             # normalise + convert, out of place
            else:
                ops.volume_normalise_(svol, ssum, split=self._split_activations())
            shyp = hyp[:, :, ea:eb].contiguous()
            st = self._regularise_and_regress(svol, shyp, None, tmp, split=self._split_activations())
            lo, hi = a - ea, b - ea
            out_rows = {k: st[k][..., lo:hi, :] for k in ("depth", "photometric_confidence", "prob_volume", "prob_volume_pre") if st[k] is not None}
        # ---- all-gather of the owners' rows: [channels, S, W] per rank, channels = depth, conf (+ 2 D probability planes) ----
        nch = 2 + (2 * D if self.return_prob_volumes else 0)
        mine = self._buffer("gather_in", (B, nch, S, W), dev, zero=True)
        if out_rows is not None:
            h = b - a
            mine[:, 0, :h] = out_rows["depth"]
            mine[:, 1, :h] = out_rows["photometric_confidence"]
            if self.return_prob_volumes:
                mine[:, 2: 2 + D, :h] = out_rows["prob_volume"]
                mine[:, 2 + D: 2 + 2 * D, :h] = out_rows["prob_volume_pre"]
        allb = self._buffer("gather_out", (world, B, nch, S, W), dev)
        if dev.type == "cuda" and hasattr(dist, "all_gather_into_tensor"):
            dist.all_gather_into_tensor(allb.view(-1), mine.view(-1), group=self.view_group)
        else:
            dist.all_gather(list(allb.unbind(0)), mine, group=self.view_group)
        self.last_collective_bytes += allb.numel() * 4
        full = allb.permute(1, 2, 0, 3, 4).reshape(B, nch, world * S, W)[:, :, :H]
        # .clone(), not .contiguous(): with world == 1 and B == 1 the slices are already contiguous VIEWS of the persistent
        # gather_out buffer and the next forward of the same shape would overwrite the results the caller holds
        res = {"depth": full[:, 0].clone(), "photometric_confidence": full[:, 1].clone(), "depth_values": depth_values,
               "prob_volume": full[:, 2: 2 + D].clone() if self.return_prob_volumes else None,
               "prob_volume_pre": full[:, 2 + D: 2 + 2 * D].clone() if self.return_prob_volumes else None}
        return res
