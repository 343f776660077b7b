"""Per-kernel timing of one cascade pass with HIP events (for bench.py's `roofline` object and DESIGN.md).

``profile_cascade`` replays exactly the launches ``CascadeDepthHead.forward`` makes, but one C-ABI call per kernel
launch, each bracketed by a pair of events recorded on torch's current stream - the stream the kernels are launched
on.  Every launch is labelled with the kernel it runs and with its ALGORITHMIC work:

  bytes   what the launch must move through HBM at least once (inputs read once + outputs written once; halo
          re-reads, weights and L2-resident re-use are NOT counted), AT THE ELEMENT SIZE THE TENSOR REALLY HAS (fp16
          activations of the "f16x2" regularisers count 2 bytes, round 4: VERDICT r3 found 4 bytes everywhere)
  flops   2 * MACs of the mathematical operator (padding taps on the volume border included, MFMA padding of
          8-channel outputs to 16 rows NOT included)

so that achieved = work / time is comparable with the 8 TB/s HBM and MFMA peaks of MI355X_MICROARCH.md.  Labels follow the
kernel symbols rocprofv3 reports (one label per template instantiation; the two warp kernels are ONE instantiation each
for all four stages), so `avg_ms` here and the average duration in profiles/*_kernel_stats.csv can be compared directly.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, List, Sequence, Tuple

import torch

from . import _lib, ops
from .cascade import CascadeDepthHead

PEAK_HBM_GBS = 8000.0          # MI355X HBM3E spec peak, MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3   # v_mfma_f32_16x16x4_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0  # dense bf16 / fp16 MFMA peak, MI355X_MICROARCH.md ("Peak BF16/FP16 MFMA ~2.5 PF dense")


def match_kernel(label: str, table: dict):
    """Key of `table` (kernel symbols as rocprofv3 prints them, profiles/pmc_traffic.json) that belongs to one of this module's launch
    labels, or None.  Labels carry the operator ("conv3d_mfma<16,16,k3,s111>"), symbols the tile configuration
    ("conv3d_mfma_bf16x3_kernel<F16Cfg<ConvCfg<16, 16, 3, 1, 1, 1, 4, 4, 16> >, false>"): operators are matched on (Cin, Cout, k, strides);
    when several tile configurations of one operator were sampled (one per cascade stage) the one with the most sampled launches wins."""
    import re
    keys = [k for k in table if not k.startswith("_")]
    if label in table:
        return label
    base = label.split(" ")[0].split("+")[0]
    cands = []
    m = re.match(r"conv3d_mfma<(\d+),(\d+),k(\d),s(\d)(\d)(\d)>", base)
    if m:
        cin, cout, kd, sd, sh, sw = m.groups()
        pat = "ConvCfg<%s, %s, %s, %s, %s, %s," % (cin, cout, kd, sd, sh, sw)
        cands = [k for k in keys if pat in k and "wgrad" not in k]
    m = re.match(r"deconv3d_mfma<(\d+),(\d+),s(\d)22>", base)
    if m:
        cin, cout, sd = m.groups()
        cands = [k for k in keys if ("DeconvCfg<%s, %s, %s," % (cin, cout, sd)) in k]
    if base == "conv3d_mfma<8,1,k3,s111>":                 # CostRegNet's 3x3x3 prob head: the Cin = 8 persistent kernel, one real output row
        cands = [k for k in keys if "ConvCfg<8, 16, 3, 1, 1, 1," in k]
    if base == "softmax_regress_kernel":                     # symbol: prob_regress_kernel<D, 0> (no fused head), one instantiation per stage
        cands = [k for k in keys if k.startswith("prob_regress_kernel<") and k.endswith(", 0>")]
    if not cands:
        stem = base.split("<")[0]
        stem = stem if stem.endswith("_kernel") else stem + "_kernel"
        cands = [k for k in keys if k == base or k.startswith(stem + "<") or k == stem or k.startswith(base + "<")]
    if not cands:
        return None
    f16 = [k for k in cands if "F16Cfg" in k]
    if f16 and len(f16) < len(cands):
        # both formats were sampled: the caller's table comes from the product default (fp16-format) run
        cands = f16
    return max(cands, key=lambda k: table[k].get("launches_sampled", 0))


def kernel_group(label: str) -> str:
    """The __global__ function a launch label belongs to, all template instantiations together: bench.py ranks kernels by THIS (round 5) -
    "conv3d_mfma<16,16,k3,s111>" and "conv3d_mfma<8,1,k3,s111> prob head" -> "conv3d_mfma_bf16x3_kernel" (the tile form and the Cin = 8
    persistent form of the same implicit-GEMM convolution), "gl_entropy_kernel<0, 2, 2, ...>" -> "gl_entropy_kernel"."""
    base = label.split(" ")[0].split("+")[0].split("<")[0]
    if base == "conv3d_mfma":
        return "conv3d_mfma_bf16x3_kernel"
    if base == "deconv3d_mfma":
        return "deconv3d_mfma_bf16x3_kernel"
    if base == "softmax_regress_kernel":
        return "prob_regress_kernel"
    return base if base.endswith("_kernel") else base + "_kernel"


def group_pmc_traffic(group: str, table: dict):
    """HBM bytes per launch of a kernel GROUP (kernel_group) from profiles/pmc_traffic.json: the launch-weighted mean over every
    symbol of that __global__ function the PMC run sampled (all tile configurations and activation formats; the forward convolution's
    tile and persistent forms together), or None."""
    stems = {"conv3d_mfma_bf16x3_kernel": ("conv3d_mfma_bf16x3_kernel<", "conv3d_mfma_bf16x3_persist_kernel<"),
             "deconv3d_mfma_bf16x3_kernel": ("deconv3d_mfma_bf16x3_kernel<", "deconv3d_mfma_bf16x3_persist_kernel<"),
             "prob_regress_kernel": ("prob_regress_kernel<",)}.get(group, (group + "<", group))
    tot, n, used = 0.0, 0, []
    for k, v in table.items():
        if k.startswith("_") or not isinstance(v, dict) or v.get("hbm_bytes_per_launch") is None:
            continue
        if any(k.startswith(st) or k == st for st in stems):
            c = v.get("launches_sampled", 0)
            tot += v["hbm_bytes_per_launch"] * c
            n += c
            used.append(k)
    return (tot / n, used) if n else (None, [])


def mfma_terms(label: str, prec: str) -> int:
    """MFMA products a kernel issues per algorithmic product: 3 (split bf16), 2 (fp16 hi + lo weights), 1 (one fp16 weight term).  "f16mix"
    (csrc/conv_kernels.hip mfma_form): one term where min(Cin, Cout) >= 32 or max >= 64 and in the visibility CNN, two elsewhere."""
    import re
    if prec in ("bf16x3", "f16x2", "f16"):
        return {"bf16x3": 3, "f16x2": 2, "f16": 1}[prec]
    if prec != "f16mix":
        return 1
    if label.startswith("vis_cnn"):
        return 1
    m = re.match(r"(?:de)?conv3d_mfma<(\d+),(\d+)", label)
    if m:
        ci, co = int(m.group(1)), int(m.group(2))
        return 1 if (min(ci, co) >= 32 or max(ci, co) >= 64) else 2
    return 2


_CUR_PREC = [None]           # conv_precision of the stage profile_cascade is timing (stages differ under the "stagemix" policy)


class Launch:
    __slots__ = ("kernel", "stage", "flops", "bytes", "start", "end", "ms", "prec")

    def __init__(self, kernel, stage, flops, nbytes):
        self.prec = _CUR_PREC[0]
        self.kernel, self.stage, self.flops, self.bytes = kernel, stage, float(flops), float(nbytes)
        self.start = torch.cuda.Event(enable_timing=True)
        self.end = torch.cuda.Event(enable_timing=True)
        self.ms = 0.0


def gather_kernel_name(which: str, code: int, C: int, D: int, W: int, tiled: bool = False, keep=None, w16=False) -> str:
    """Symbol rocprofv3 reports for the gather pass of a stage (template args: feature dtype, C / 8, work-items per pixel,
    octet-tiled layout; the entropy pass also: does it keep the per-view correlations; fp16 window - a keeping pass with fp32 windows
    is the exact form, MVS_CORR_F32)."""
    import os
    w16 = bool(w16) and (bool(keep) or not os.environ.get("MVS_GATHER_WINDOW", "").startswith("f3"))
    b = lambda v: "true" if v else "false"
    if C in (8, 16, 32, 64) and W % 8 == 0:
        nch = (D + 3) // 4
        ns = 8 if nch >= 8 else 4 if nch >= 4 else 2 if nch >= 2 else 1
        if which == "entropy":
            return "gl_entropy_kernel<%d, %d, %d, %s, %s, %s>" % (code, C // 8, ns, b(tiled), b(keep), b(w16))
        return "gl_%s_kernel<%d, %d, %d, %s, %s>" % (which, code, C // 8, ns, b(tiled), b(w16))
    return "warp_corr_%s_kernel" % which


def _conv_name(cin, cout, kd, stride):
    return "conv3d_mfma<%d,%d,k%d,s%d%d%d>" % (cin, cout, kd, stride[0], stride[1], stride[2])


def _timed(launches: List[Launch], kernel: str, stage: int, flops: float, nbytes: float, fn):
    rec = Launch(kernel, stage, flops, nbytes)
    rec.start.record()
    out = fn()
    rec.end.record()
    launches.append(rec)
    return out


def _regnet_layers(net, vol, stage, launches, precision, fused_head=False, split=False):
    """The nine U-Net launches of mvs_regnet_fwd, one C-ABI call each (same kernels, same order).  fused_head: the last layer
    carries the 1x1x1 `prob` head (mvs_regnet_logits_fwd) and returns logits instead of features.  split: the activations (`vol`
    included) are in the split format of MVS_PREC_BF16X3_SPLIT, as in StageNet's inference path."""
    ws, bs, prob_w, prob_b = net.packed_all(vol.device, precision)
    prec = _lib.PREC_BF16X3_SPLIT if split else _lib.PRECISIONS[precision]
    three_d = net.kind == _lib.REG_COSTREGNET3D
    s2 = (1, 2, 2) if three_d else (2, 2, 2)
    sd = 1 if three_d else 2

    def conv(x, i, cout, stride):
        B, D, H, W, cin = x.shape
        od = (D - 1) // stride[0] + 1
        oh, ow = (H - 1) // stride[1] + 1, (W - 1) // stride[2] + 1
        nout = B * od * oh * ow
        return _timed(launches, _conv_name(cin, cout, 3, stride), stage, 2.0 * 27 * cin * cout * nout,
                      float(x.element_size()) * (x.numel() + nout * cout), lambda: ops.conv3d_bn_relu(x, ws[i], bs[i], cout, 3, stride, True, prec))

    def deconv(x, i, cout, skip):
        B, D, H, W, cin = x.shape
        nout = B * D * sd * 4 * H * W
        return _timed(launches, "deconv3d_mfma<%d,%d,s%d22>" % (cin, cout, sd), stage, 2.0 * 27 * cin * cout * (B * D * H * W),
                      float(x.element_size()) * (x.numel() + 2 * nout * cout), lambda: ops.deconv3d_bn_relu_add(x, ws[i], bs[i], cout, sd, skip, prec))

    c1 = conv(vol, 0, 16, s2)
    c2 = conv(c1, 1, 16, (1, 1, 1))
    c3 = conv(c2, 2, 32, s2)
    c4 = conv(c3, 3, 32, (1, 1, 1))
    c5 = conv(c4, 4, 64, s2)
    c6 = conv(c5, 5, 64, (1, 1, 1))
    x = deconv(c6, 6, 32, c4)
    x = deconv(x, 7, 16, c2)
    if fused_head:
        B, D, H, W, cin = x.shape
        nout = B * D * sd * 4 * H * W
        # input + skip (the cost volume) in the activation format, logits out in fp32
        return _timed(launches, "deconv3d_mfma<%d,%d,s%d22>+prob" % (cin, 8, sd), stage, 2.0 * 27 * cin * 8 * (B * D * H * W) + 2.0 * 8 * nout,
                      float(x.element_size()) * (x.numel() + nout * 8) + 4.0 * nout, lambda: ops.deconv3d_prob(x, ws[8], bs[8], sd, vol, prob_w, prob_b, prec))
    return deconv(x, 8, 8, vol)


def _transformer_layers(net, vol, pos, stage, launches):
    """The launches of PureTransformerCostReg.logits_cl one by one (same kernels, same order)."""
    import math
    P = net._cache.get(net, net._build, "bf16x3")
    prec = _lib.PREC_BF16X3
    B, D, H, W, _ = vol.shape
    rate = net.rate
    n = (D // rate[0]) * (H // rate[1]) * (W // rate[2])
    kp = 8 * rate[0] * rate[1] * rate[2]
    T = B * n
    x = _timed(launches, "tr_gemm<embed>", stage, 2.0 * T * kp * 64, 4.0 * (vol.numel() + (3 * B * D * H * W if pos is not None else 0) + T * 64),
               lambda: ops.tr_embed(vol, pos, P["pe_w"], P["pe_div"], P["down_w"], P["down_b"], *P["down_ln"], rate, prec))
    scale = (64 // net.num_heads) ** -0.5
    if net.softmax_scale == "entropy_invariance":
        scale *= math.log(n, net.train_avg_length)
    for L in P["layers"]:
        a = _timed(launches, "[bundle] tr_gemm<qkv>+tr_attention", stage, 2.0 * T * 64 * 192 + 4.0 * B * n * n * 64,
                   4.0 * T * (64 + 64) + 2.0 * T * 192 * 2, lambda: ops.tr_attention(x, L["qkv"], net.num_heads, scale, prec, net.attention_code()))
        x = _timed(launches, "tr_gemm<res_ln,64>", stage, 2.0 * T * 64 * 64, 4.0 * T * 64 * 3,
                   lambda: ops.tr_linear(a, L["proj"], L["proj_b"], _lib.TR_EPI_RES_LN, 64, prec, residual=x, gamma=L["g1"],
                                         ln_w=L["n1"][0], ln_b=L["n1"][1], ln_eps=L["n1"][2]))
        hdn = _timed(launches, "tr_gemm<gelu>", stage, 2.0 * T * 64 * 256, 4.0 * T * (64 + 256),
                     lambda: ops.tr_linear(x, L["l1"], L["l1_b"], _lib.TR_EPI_GELU, 256, prec))
        x = _timed(launches, "tr_gemm<res_ln,256>", stage, 2.0 * T * 256 * 64, 4.0 * T * (256 + 128),
                   lambda: ops.tr_linear(hdn, L["l2"], L["l2_b"], _lib.TR_EPI_RES_LN, 64, prec, residual=x, gamma=L["g2"],
                                         ln_w=L["n2"][0], ln_b=L["n2"][1], ln_eps=L["n2"][2]))
    return _timed(launches, "tr_gemm<up>", stage, 2.0 * T * 64 * kp, 4.0 * (T * 64 + B * D * H * W),
                  lambda: ops.tr_up_prob(x, P["up_w"], P["up_b"], *P["up_ln"], P["prob_w"], P["prob_b"], (D, H, W), rate, prec))


@torch.no_grad()
def profile_cascade(head: CascadeDepthHead, features, proj_matrices, depth_values, tmp=(5.0, 5.0, 5.0, 1.0)) -> Tuple[dict, List[Launch]]:
    launches: List[Launch] = []
    n = len(head.ndepths)
    out = None
    confs = []
    final_conf = None
    pe_range = None
    # Round 5: let the host run AHEAD of the device for the whole pass.  Issuing one reference view's ~60 launches + 120 events from Python takes
    # longer (~2 ms) than the device needs for the short launches, so an idle device used to wait for the host between a start event and its
    # kernel - the "~8 us of event overhead" rounds 3-4 reported on every sub-20-us launch (bench.py's avg_launch_ms 28 us against rocprofv3's
    # 23 us for the same convolutions).  A spin kernel at the head of the stream (torch.cuda._sleep, ~10 ms) lets the whole pass queue up behind
    # it: the events then bracket back-to-back device work.
    if depth_values.is_cuda:
        torch.cuda._sleep(int(2.0e7))
    # round 5: ONE prologue launch (every stage's homographies + stage 1's hypotheses), exactly as CascadeDepthHead.forward issues it
    H0, W0 = features["stage1"].shape[-2:]
    B0 = depth_values.shape[0]
    homs, hyp0 = _timed(launches, "cascade_prologue", 0, 0, 4.0 * B0 * head.ndepths[0] * H0 * W0,
                        lambda: ops.cascade_prologue([proj_matrices["stage%d" % (i + 1)] for i in range(n)], depth_values, head.ndepths[0], H0, W0,
                                                     inverse=head.inverse_depth))
    for s in range(n):
        key = "stage%d" % (s + 1)
        net = head.fusions[s]
        _CUR_PREC[0] = net.conv_precision
        feats, code = ops._feat(features[key])
        proj = proj_matrices[key]
        B, V, C, H, W = feats.shape
        D = head.ndepths[s]
        HW = H * W
        esz = feats.element_size()
        if s == 0:
            hyp = hyp0
        elif out.get("next_hyp") is not None:
            hyp = out["next_hyp"]                            # scheduled by the previous stage's head (round 5)
        else:
            pd, ph = out["depth"], out["depth_values"]
            hyp = _timed(launches, "schedule_range", s, 0, 4.0 * (B * D * HW + 3 * B * HW / 4),
                         lambda: ops.schedule_inverse_range(pd, ph, D, head.depth_interals_ratio[s], H, W))
        hom = homs[s]
        last = s == n - 1                                   # the last stage's head also writes the cascade's averaged confidence (a16 fused)
        cprev = list(confs) if last else None
        # ... and every other stage's head the next stage's hypotheses (a14 fused), exactly when CascadeDepthHead.forward asks for it
        sched = (not last and head.inverse_depth and D >= 3 and tuple(features["stage%d" % (s + 2)].shape[-2:]) == (2 * H, 2 * W))
        Dn = head.ndepths[s + 1] if not last else 0
        rn = head.depth_interals_ratio[s + 1] if not last else 0.0

        def run_head(lg):
            if sched:
                return ops.softmax_regress_schedule(lg, hyp, tmp[s], _lib.HEAD_CE_EVAL, 0, net.return_prob_volumes, Dn, rn)
            return ops.softmax_regress(lg, hyp, tmp[s], _lib.HEAD_CE_EVAL, 0, net.return_prob_volumes, conf_prev=cprev)
        head_label = "prob_regress_sched_kernel" if sched else "softmax_regress_kernel"
        head_bytes = 4.0 * B * ((2 + int(bool(net.return_prob_volumes))) * D * HW + 2 * HW + (HW if last else 0) + (4 * Dn * HW if sched else 0))
        corr_flops = 2.0 * (V - 1) * B * D * HW * C * 5          # 4-tap bilinear + correlation MAC per channel
        # every feature map once, the hypotheses once, the entropy maps out
        tiled = isinstance(feats, ops.PackedFeatures)
        keep = net._keeps_correlations(feats, 8, hyp)                                   # exactly StageNet.forward's choice
        w16 = net.gather_precision == "f16"                                             # fp16 source windows (+ fp16 kept correlations)
        corr_bytes = B * (V - 1) * D * HW * (16.0 if w16 else 32.0)                     # per-view group correlations, fp16 / fp32 (as-built traffic)
        if keep:
            ent, corr = _timed(launches, gather_kernel_name("entropy", code, C, D, W, tiled, True, w16), s, corr_flops,
                               B * (V * C * HW * esz + D * HW * 4 + (V - 1) * HW * 4) + corr_bytes,
                               lambda: ops.warp_corr_entropy_keep(feats, code, hom, hyp, 8, exact=not w16))
        else:
            ent = _timed(launches, gather_kernel_name("entropy", code, C, D, W, tiled, False, w16), s, corr_flops, B * (V * C * HW * esz + D * HW * 4 + (V - 1) * HW * 4),
                         lambda: ops.warp_corr_entropy(feats, code, hom, hyp, 8, f16_window=w16))
        vp = net._vis_params(feats.device)
        prec = _lib.PRECISIONS[net._vis_precision()]
        N = B * (V - 1)
        vis_flops = 2.0 * N * HW * (9 * 16 + 9 * 16 * 16 + 9 * 16 * 8 + 8)
        if net._vis_precision() in ("bf16x3",) + _lib.F16_FORMATS:
            # one row-streaming launch; algorithmic traffic = entropy in + visibility out
            vis = _timed(launches, "vis_cnn_kernel", s, vis_flops, 4.0 * N * HW * 2, lambda: ops.vis_weight(ent, vp, prec))
        else:
            # fp32 mode: the four launches of mvs_vis_weight_fwd one by one
            t1 = _timed(launches, "vis_conv1", s, 2.0 * N * HW * 9 * 16, 4.0 * N * HW * 17, lambda: ops.vis_conv1(ent, vp[0], vp[1]))
            t1 = t1.reshape(1, N, H, W, 16)
            t2 = _timed(launches, _conv_name(16, 16, 1, (1, 1, 1)), s, 2.0 * N * HW * 9 * 16 * 16, 4.0 * N * HW * 32,
                        lambda: ops.conv3d_bn_relu(t1, vp[2], vp[3], 16, 1, (1, 1, 1), True, prec))
            t3 = _timed(launches, _conv_name(16, 8, 1, (1, 1, 1)), s, 2.0 * N * HW * 9 * 16 * 8, 4.0 * N * HW * 24,
                        lambda: ops.conv3d_bn_relu(t2, vp[4], vp[5], 8, 1, (1, 1, 1), True, prec))
            vis = _timed(launches, "vis_out", s, 2.0 * N * HW * 8, 4.0 * N * HW * 9,
                         lambda: ops.vis_out(t3.reshape(N, H, W, 8), vp[6], vp[7], ent.shape))
        split = net._split_activations()
        f16 = net._f16_activations()
        agg_name = gather_kernel_name("aggregate", code, C, D, W, tiled, w16=f16 and ops.gather_is_lds_staged(feats, 8, hyp))
        if keep:
            vol = _timed(launches, "corr_aggregate_kernel", s, 2.0 * B * (V - 1) * D * HW * 8, corr_bytes + B * ((V - 1) * HW * 4 + 8 * D * HW * (2 if f16 else 4)),
                         lambda: ops.corr_aggregate(corr, vis, split=split, f16=f16))
            del corr
        elif f16 and not ops.gather_is_lds_staged(feats, 8, hyp):
            # shapes outside the LDS-staged gather: fp32 volume + conversion, exactly as StageNet.forward does
            v32 = _timed(launches, agg_name, s, corr_flops, B * (V * C * HW * esz + D * HW * 4 + (V - 1) * HW * 4 + 8 * D * HW * 4),
                         lambda: ops.warp_corr_aggregate(feats, code, hom, hyp, vis, 8, normalise=True)[0])
            vol = _timed(launches, "volume_to_f16", s, 0, B * 8 * D * HW * 6.0, lambda: ops.volume_to_f16(v32))
        else:
            vol = _timed(launches, agg_name, s, corr_flops,
                         B * (V * C * HW * esz + D * HW * 4 + (V - 1) * HW * 4 + 8 * D * HW * (2 if f16 else 4)),
                         lambda: ops.warp_corr_aggregate(feats, code, hom, hyp, vis, 8, split=split, f16=f16)[0])
        if getattr(net.cost_reg, "kind", None) == "transformer":
            pos = None
            if head.use_pe3d:
                pr = pe_range
                pos, pe_range = _timed(launches, "[bundle] pos3d", s, 0, 4.0 * 4 * B * D * HW,
                                       lambda: ops.position3d(proj[:, 0, 1, :3, :3], hyp, depth_values, pr))
            logits = _transformer_layers(net.cost_reg, vol, pos, s, launches)
            r3 = _timed(launches, head_label, s, 0, head_bytes, lambda: run_head(logits))
            out = {"depth": r3[0], "photometric_confidence": r3[1], "depth_values": hyp, "next_hyp": r3[3] if sched else None}
            confs.append(r3[1])
            if last:
                final_conf = r3[3]
            continue
        ks = net.cost_reg.prob_ksize
        if ks == 1 and net.conv_precision in _lib.F16_FORMATS + ("bf16x3",) and net.fuse_prob_head:
            logits = _regnet_layers(net.cost_reg, vol, s, launches, net.conv_precision, fused_head=True, split=split)
            r3 = _timed(launches, head_label, s, 0, head_bytes, lambda: run_head(logits))
            out = {"depth": r3[0], "photometric_confidence": r3[1], "depth_values": hyp, "next_hyp": r3[3] if sched else None}
            confs.append(r3[1])
            if last:
                final_conf = r3[3]
            continue
        feat_cl = _regnet_layers(net.cost_reg, vol, s, launches, net.conv_precision, split=split)
        ws, bs, prob_w, prob_b = net.cost_reg.packed_all(feats.device, net.conv_precision)
        if ks == 3 and net.conv_precision in _lib.F16_FORMATS + ("bf16x3",):
            logits = _timed(launches, "conv3d_mfma<8,1,k3,s111> prob head", s, 2.0 * B * D * HW * 8 * 27, B * (float(feat_cl.element_size()) * 8 * D * HW + 4.0 * D * HW),
                            lambda: ops.conv3d_logits(feat_cl, prob_w, prob_b, _lib.PREC_BF16X3_SPLIT if split else _lib.PRECISIONS[net.conv_precision]))
            r3 = _timed(launches, head_label, s, 0, head_bytes, lambda: run_head(logits))
            out = {"depth": r3[0], "photometric_confidence": r3[1], "depth_values": hyp, "next_hyp": r3[3] if sched else None}
            confs.append(r3[1])
            if last:
                final_conf = r3[3]
            continue
        res = _timed(launches, "prob_regress_kernel<%d,%d>" % (D, ks), s, 2.0 * B * D * HW * 8 * (27 if ks == 3 else 1),
                     4.0 * B * (8 * D * HW + D * HW + 2 * D * HW + 2 * HW),
                     lambda: ops.prob_regress(feat_cl, prob_w, prob_b, ks, hyp, tmp[s], _lib.HEAD_CE_EVAL, 0, net.return_prob_volumes))
        out = {"depth": res[0], "photometric_confidence": res[1], "depth_values": hyp}
        confs.append(res[1])
    Hf, Wf = features["stage%d" % n].shape[-2:]
    if final_conf is None:                                  # fp32-exact route: no fused form
        final_conf = _timed(launches, "confidence_average", n - 1, 0, 4.0 * Hf * Wf * 2, lambda: ops.confidence_average(confs, Hf, Wf))
    torch.cuda.synchronize()
    for l in launches:
        l.ms = l.start.elapsed_time(l.end)
    return {"refined_depth": out["depth"], "photometric_confidence": final_conf}, launches


def summarize(launches: Sequence[Launch]) -> "OrderedDict[str, dict]":
    """Aggregate by kernel name: calls, total ms, average ms, achieved GB/s and TFLOP/s."""
    agg: "OrderedDict[str, dict]" = OrderedDict()
    for l in launches:
        a = agg.setdefault(l.kernel, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "issued_flops": 0.0})
        a["calls"] += 1
        a["ms"] += l.ms
        a["flops"] += l.flops
        a["bytes"] += l.bytes
        a["issued_flops"] += l.flops * (mfma_terms(l.kernel, l.prec) if l.prec else 1)       # MFMA products actually issued (terms per product)
    for a in agg.values():
        t = max(a["ms"], 1e-9) * 1e-3
        a["avg_ms"] = a["ms"] / a["calls"]
        a["gbs"] = a["bytes"] / t / 1e9
        a["tflops"] = a["flops"] / t / 1e12
    return agg
