"""Tensor-level wrappers over the C ABI (include/mvs_hip.h).  Each function allocates its outputs with torch
(so the caching allocator owns them), enqueues the HIP kernels on the caller's current stream and returns
fresh tensors.  No autograd: this is the inference path (SURVEY.md section 8f #2 lists backward as next).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import check, lib, ptr, stream_of


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.contiguous() if t.dtype == torch.float32 else t.float().contiguous()


class PackedFeatures:
    """Feature maps of one stage in the hand-off layout: ``data`` [B,V,C/8,H,W,8], fp32 / bf16 / fp16.
    Quacks like the [B,V,C,H,W] tensor the reference passes (``shape``, ``device``, ``dtype``) so StageNet.forward accepts either."""

    def __init__(self, data: torch.Tensor):
        assert data.dim() == 6 and data.shape[-1] == 8 and data.is_contiguous(), "PackedFeatures wants a dense [B,V,C/8,H,W,8] tensor"
        self.data = data

    @property
    def shape(self):
        B, V, O, H, W, _ = self.data.shape
        return torch.Size((B, V, O * 8, H, W))

    @property
    def device(self):
        return self.data.device

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def requires_grad(self):
        return False

    def element_size(self):
        return self.data.element_size()

    def to(self, *a, **k):
        return PackedFeatures(self.data.to(*a, **k).contiguous())

    def unpack(self) -> torch.Tensor:
        """Back to planar [B,V,C,H,W] (tests / debugging)."""
        B, V, O, H, W, _ = self.data.shape
        return self.data.permute(0, 1, 2, 5, 3, 4).reshape(B, V, O * 8, H, W).contiguous()


def pack_features(features: torch.Tensor, dtype: Optional[torch.dtype] = None) -> PackedFeatures:
    """[B,V,C,H,W] (fp32 / bf16 / fp16) -> PackedFeatures in `dtype` (default: keep).  One read + one write of the features;
    a producer that emits the tiled layout itself has nothing to pack."""
    f, code = _feat(features)
    B, V, Cc, H, W = f.shape
    dtype = f.dtype if dtype is None else dtype
    out = torch.empty(B, V, Cc // 8, H, W, 8, dtype=dtype, device=f.device)
    check(lib().mvs_pack_features(ptr(f), code, ptr(out), _lib.DTYPE_CODE[dtype], B * V, Cc, H, W, stream_of(f)), "mvs_pack_features")
    return PackedFeatures(out)


def feature_conv_is_built(cin: int, cout: int) -> bool:
    return bool(lib().mvs_feature_conv_is_built(int(cin), int(cout)))


def conv2d3x3_tiles(x: torch.Tensor, w_packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, swish: bool = False,
                    out: Optional[torch.Tensor] = None, dtype: torch.dtype = torch.bfloat16) -> torch.Tensor:
    """Producer-side feature emitter: x [N,Cin,H,W] planar (fp32 / bf16 / fp16; any batch stride) through
    Conv2d(Cin, cout, 3, padding=1) [+ bias] [+ Swish] -> octet tiles [N, cout/8, H, W, 8] in `dtype`, written by the convolution's
    epilogue (no planar output, no pack pass).  `out`: a [N, cout/8, H, W, 8] tensor or VIEW to fill - e.g. `packed.data[:, v]` of a
    [B, V, cout/8, H, W, 8] PackedFeatures buffer (any batch stride, the rest contiguous)."""
    if x.dtype not in _lib.DTYPE_CODE:
        x = x.float()
    N, cin, H, W = x.shape
    if x.stride()[1:] != (H * W, W, 1):
        x = x.contiguous()
    if out is None:
        out = torch.empty(N, cout // 8, H, W, 8, dtype=dtype, device=x.device)
    assert out.shape == (N, cout // 8, H, W, 8) and out.stride()[1:] == (H * W * 8, W * 8, 8, 1) and out.dtype in _lib.DTYPE_CODE, "tiled output [N,C/8,H,W,8]"
    check(lib().mvs_conv2d3x3_tiles_fwd(ptr(x), _lib.DTYPE_CODE[x.dtype], ptr(w_packed), ptr(bias), 1 if swish else 0, ptr(out), _lib.DTYPE_CODE[out.dtype],
                                        N, cin, cout, H, W, x.stride(0) if N > 1 else cin * H * W, out.stride(0) if N > 1 else cout * H * W,
                                        stream_of(x)), "mvs_conv2d3x3_tiles_fwd")
    return out


def _feat(t) -> Tuple[torch.Tensor, int]:
    if isinstance(t, PackedFeatures):
        return t, _lib.DTYPE_CODE[t.dtype]
    if t.dtype not in _lib.DTYPE_CODE:
        t = t.float()
    return t.contiguous(), _lib.DTYPE_CODE[t.dtype]


def _feat_ptr(features):
    """(tensor handed to the C ABI, layout code)"""
    if isinstance(features, PackedFeatures):
        return features.data, _lib.LAYOUT_OCTET_TILED
    return features, _lib.LAYOUT_PLANAR


# ---- a1 + warping.py:80 -------------------------------------------------------------------------
def compose_homography(proj_matrices: torch.Tensor) -> torch.Tensor:
    """proj_matrices [B,V,2,4,4] -> per-source-view homographies [B,V-1,12] (R row-major, then t)."""
    p = _f32c(proj_matrices)
    B, V = p.shape[:2]
    assert p.shape[2:] == (2, 4, 4), "proj_matrices must be [B,V,2,4,4]"
    out = torch.empty(B, V - 1, 12, dtype=torch.float32, device=p.device)
    check(lib().mvs_compose_homography(ptr(p), B, V, ptr(out), stream_of(p)), "mvs_compose_homography")
    return out


def cascade_prologue(proj_matrices: Sequence[torch.Tensor], depth_values: Optional[torch.Tensor] = None, ndepths: int = 0, H: int = 0, W: int = 0,
                     inverse: bool = True):
    """Round 5: the cascade's prologue in ONE launch (mvs_cascade_prologue_fwd): the homographies of every stage's proj_matrices
    [B,V,2,4,4] -> list of [B,V-1,12], and - depth_values [B,N] given - stage 1's hypotheses [B,ndepths,H,W] (init_range)."""
    ps = [_f32c(p) for p in proj_matrices]
    B, V = ps[0].shape[:2]
    for p in ps:
        assert p.shape == (B, V, 2, 4, 4), "proj_matrices must be [B,V,2,4,4] with the same B, V on every stage"
    hom = torch.empty(len(ps), B, V - 1, 12, dtype=torch.float32, device=ps[0].device)
    hyp, dv, N = None, None, 0
    if depth_values is not None:
        dv = _f32c(depth_values)
        assert dv.dim() == 2 and dv.shape[0] == B, "depth_values must be [B,N] (per-pixel ranges go through init_range)"
        N = dv.shape[1]
        hyp = torch.empty(B, ndepths, H, W, dtype=torch.float32, device=dv.device)
    pa = _ptr_array(ps)
    check(lib().mvs_cascade_prologue_fwd(C.cast(pa, C.c_void_p), len(ps), B, V, ptr(hom), ptr(dv), N, 1 if inverse else 0, ptr(hyp), ndepths, H, W,
                                         stream_of(ps[0])), "mvs_cascade_prologue_fwd")
    return list(hom.unbind(0)), hyp


def homography_from_proj(src_proj: torch.Tensor, ref_proj: torch.Tensor) -> torch.Tensor:
    s, r = _f32c(src_proj), _f32c(ref_proj)
    B = s.shape[0]
    out = torch.empty(B, 12, dtype=torch.float32, device=s.device)
    check(lib().mvs_homography_from_proj(ptr(s), ptr(r), B, ptr(out), stream_of(s)), "mvs_homography_from_proj")
    return out


def homo_warp(src_fea: torch.Tensor, homography: torch.Tensor, depth_values: torch.Tensor,
              want_warped: bool = True, want_mask: bool = True):
    f, code = _feat(src_fea)
    B, Cc, H, W = f.shape
    dv = _f32c(depth_values)
    D = dv.shape[1]
    is_vol = 1 if dv.dim() == 4 else 0
    warped = torch.empty(B, Cc, D, H, W, dtype=torch.float32, device=f.device) if want_warped else None
    mask = torch.empty(B, D, H, W, dtype=torch.uint8, device=f.device) if want_mask else None
    check(lib().mvs_homo_warp_fwd(ptr(f), code, ptr(homography.contiguous()), ptr(dv), is_vol, ptr(warped), ptr(mask),
                                  B, Cc, D, H, W, stream_of(f)), "mvs_homo_warp_fwd")
    return warped, (mask.bool() if mask is not None else None)


# ---- a2-a6 --------------------------------------------------------------------------------------
def warp_corr_entropy(features: torch.Tensor, code: int, homography: torch.Tensor, hyp: torch.Tensor, G: int,
                      view_begin: int = 1, view_end: Optional[int] = None, out: Optional[torch.Tensor] = None, f16_window: bool = False) -> torch.Tensor:
    """features [B,V,C,H,W] contiguous; -> entropy [B,V-1,H,W] (only views in [view_begin, view_end) are written; pass a
    preallocated `out` to skip the zero fill when every view is written).  f16_window: MVS_GATHER_F16 (the fp16 formats' gather:
    source window staged as fp16)."""
    B, V, Cc, H, W = features.shape
    D = hyp.shape[1]
    view_end = V if view_end is None else view_end
    if out is not None:
        ent = out
    elif view_begin == 1 and view_end == V:
        ent = torch.empty(B, V - 1, H, W, dtype=torch.float32, device=features.device)
    else:
        ent = torch.zeros(B, V - 1, H, W, dtype=torch.float32, device=features.device)
    ft, layout = _feat_ptr(features)
    check(lib().mvs_warp_corr_entropy_fwd(ptr(ft), code, layout, ptr(homography), ptr(hyp), ptr(ent), B, V, Cc, G, D, H, W,
                                          view_begin, view_end, 1 if f16_window else 0, stream_of(ft)), "mvs_warp_corr_entropy_fwd")
    return ent


def vis_weight(entropy: torch.Tensor, params: Sequence[torch.Tensor], precision: int = 0) -> torch.Tensor:
    """entropy [..., H, W] -> sigmoid(vis CNN) of the same shape.  params = packed (w1,b1,w2,b2,w3,b3,w4,b4)."""
    e = _f32c(entropy)
    H, W = e.shape[-2:]
    N = e.numel() // (H * W)
    vis = torch.empty_like(e)
    nbytes = lib().mvs_vis_workspace_bytes(N, H, W, precision)
    ws = torch.empty(max(nbytes // 4, 4), dtype=torch.float32, device=e.device)
    check(lib().mvs_vis_weight_fwd(ptr(e), *[ptr(p) for p in params], ptr(vis), ptr(ws), nbytes, N, H, W, precision, stream_of(e)),
          "mvs_vis_weight_fwd")
    return vis


def vis_conv1(entropy: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor) -> torch.Tensor:
    e = _f32c(entropy)
    H, W = e.shape[-2:]
    N = e.numel() // (H * W)
    out = torch.empty(N, H, W, 16, dtype=torch.float32, device=e.device)
    check(lib().mvs_vis_conv1_fwd(ptr(e), ptr(w1), ptr(b1), ptr(out), N, H, W, stream_of(e)), "mvs_vis_conv1_fwd")
    return out


def vis_out(x_cl8: torch.Tensor, w4: torch.Tensor, b4: torch.Tensor, shape) -> torch.Tensor:
    N, H, W, _ = x_cl8.shape
    vis = torch.empty(shape, dtype=torch.float32, device=x_cl8.device)
    check(lib().mvs_vis_out_fwd(ptr(x_cl8), ptr(w4), ptr(b4), ptr(vis), N, H, W, stream_of(x_cl8)), "mvs_vis_out_fwd")
    return vis


def warp_corr_aggregate(features: torch.Tensor, code: int, homography: torch.Tensor, hyp: torch.Tensor, vis: torch.Tensor,
                        G: int, normalise: bool = True, view_begin: int = 1, view_end: Optional[int] = None, out=None, split: bool = False,
                        f16: bool = False):
    """-> (volume_cl [B,D,H,W,G], vis_sum [B,H,W] or None).  `out` = preallocated (volume, vis_sum) to fill.
    split: leave the (normalised, G = 8) volume in the split activation format of the bf16x3 U-Net (to_split / from_split);
    f16: write it as fp16 (the cost volume of the MVS_PREC_F16X2 U-Net; LDS-staged gather shapes only)."""
    B, V, Cc, H, W = features.shape
    D = hyp.shape[1]
    view_end = V if view_end is None else view_end
    if out is not None:
        vol, vsum = out
    else:
        vol = torch.empty(B, D, H, W, G, dtype=torch.float16 if f16 else torch.float32, device=features.device)
        vsum = None if normalise else torch.empty(B, H, W, dtype=torch.float32, device=features.device)
    ft, layout = _feat_ptr(features)
    check(lib().mvs_warp_corr_aggregate_fwd(ptr(ft), code, layout, ptr(homography), ptr(hyp), ptr(vis), ptr(vol), ptr(vsum),
                                            1 if normalise else 0, _lib.VOLUME_F16 if f16 else (_lib.VOLUME_SPLIT if split else _lib.VOLUME_F32), B, V, Cc, G, D, H, W,
                                            view_begin, view_end, stream_of(ft)), "mvs_warp_corr_aggregate_fwd")
    return vol, vsum


def warp_corr_aggregate_bwd(features: torch.Tensor, code: int, homography: torch.Tensor, hyp: torch.Tensor, vis: torch.Tensor,
                            vis_sum: torch.Tensor, volume_cl: torch.Tensor, grad_volume_cl: torch.Tensor, G: int):
    """Backward of warp_corr_aggregate(normalise=True) -> (grad_features [B,V,C,H,W] fp32, grad_vis [B,V-1,H,W])."""
    B, V, Cc, H, W = features.shape
    D = hyp.shape[1]
    gfeat = torch.empty(B, V, Cc, H, W, dtype=torch.float32, device=features.device)
    gvis = torch.empty(B, V - 1, H, W, dtype=torch.float32, device=features.device)
    check(lib().mvs_warp_corr_aggregate_bwd(ptr(features), code, ptr(homography), ptr(hyp), ptr(vis), ptr(vis_sum), ptr(volume_cl),
                                            ptr(grad_volume_cl), ptr(gfeat), ptr(gvis), B, V, Cc, G, D, H, W, stream_of(features)),
          "mvs_warp_corr_aggregate_bwd")
    return gfeat, gvis


def volume_normalise_(vol_cl: torch.Tensor, vis_sum: torch.Tensor, split: bool = False) -> torch.Tensor:
    B, D, H, W, G = vol_cl.shape
    check(lib().mvs_volume_normalise(ptr(vol_cl), ptr(vis_sum), B, D, H, W, G, _lib.VOLUME_SPLIT if split else _lib.VOLUME_F32,
                                     stream_of(vol_cl)), "mvs_volume_normalise")
    return vol_cl


def gather_is_lds_staged(features, G: int, hyp: torch.Tensor) -> bool:
    """Do the gather passes take the LDS-staged kernels for this call (they alone write the split / fp16 volume directly)?"""
    B, V, Cc, H, W = features.shape
    _, layout = _feat_ptr(features)
    return bool(lib().mvs_gather_is_lds_staged(layout, Cc, G, hyp.shape[1], H, W))


def gather_keeps_correlations(features, G: int, hyp: torch.Tensor) -> bool:
    """Is the correlation-keeping pass 1 (warp_corr_entropy_keep + corr_aggregate) built for this call's shapes?"""
    B, V, Cc, H, W = features.shape
    _, layout = _feat_ptr(features)
    return bool(lib().mvs_gather_keeps_correlations(layout, Cc, G, hyp.shape[1], H, W))


def warp_corr_entropy_keep(features, code: int, homography: torch.Tensor, hyp: torch.Tensor, G: int, exact: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """Pass 1 over all source views that also KEEPS the per-view group correlations (cost_volume.py:79-84):
    -> (entropy [B,V-1,H,W] fp32, corr [B,V-1,D,H,W,8]).  corr is fp16 from fp16 source windows (MVS_CORR_F16), or - `exact` - fp32 from
    fp32 windows (MVS_CORR_F32): corr_aggregate then writes what the second gather would have."""
    B, V, Cc, H, W = features.shape
    D = hyp.shape[1]
    ft, layout = _feat_ptr(features)
    ent = torch.empty(B, V - 1, H, W, dtype=torch.float32, device=ft.device)
    corr = torch.empty(B, V - 1, D, H, W, 8, dtype=torch.float32 if exact else torch.float16, device=ft.device)
    check(lib().mvs_warp_corr_entropy_keep_fwd(ptr(ft), code, layout, ptr(homography), ptr(hyp), ptr(ent), ptr(corr),
                                               _lib.CORR_F32 if exact else _lib.CORR_F16, B, V, Cc, G, D, H, W, stream_of(ft)),
          "mvs_warp_corr_entropy_keep_fwd")
    return ent, corr


def corr_aggregate(corr: torch.Tensor, vis: torch.Tensor, split: bool = False, f16: bool = True) -> torch.Tensor:
    """corr [B,V-1,D,H,W,8] fp16 or fp32 (warp_corr_entropy_keep), vis [B,V-1,H,W] fp32 -> the normalised cost volume [B,D,H,W,8]
    (cost_volume.py:97-101): fp16 (f16, the fp16 U-Nets' format), the split activation format of the bf16x3 U-Net (split) or fp32
    (neither; the transformer regulariser)."""
    assert corr.dtype in (torch.float16, torch.float32) and corr.is_contiguous() and corr.dim() == 6 and corr.shape[-1] == 8
    B, NV, D, H, W, _ = corr.shape
    v = _f32c(vis)
    assert v.shape == (B, NV, H, W)
    f16 = f16 and not split
    vol = torch.empty(B, D, H, W, 8, dtype=torch.float16 if f16 else torch.float32, device=corr.device)
    fmt = _lib.VOLUME_F16 if f16 else (_lib.VOLUME_SPLIT if split else _lib.VOLUME_F32)
    cfmt = _lib.CORR_F32 if corr.dtype == torch.float32 else _lib.CORR_F16
    check(lib().mvs_corr_aggregate_fwd(ptr(corr), cfmt, ptr(v), ptr(vol), fmt, B, NV + 1, D, H, W, stream_of(corr)), "mvs_corr_aggregate_fwd")
    return vol


def f16_saturation_count(reset: bool = False, device=None) -> int:
    """Work-items that stored an fp16 ACTIVATION beyond +-65504 (clamped) on `device` (default: the current one) since the last reset -
    0 unless the "f16x2" format degraded something the fp32-equivalent "bf16x3" would have kept.  Synchronises the device."""
    if device is not None and torch.device(device).type == "cuda":
        with torch.cuda.device(device):
            return int(lib().mvs_f16_saturation_count(1 if reset else 0))
    return int(lib().mvs_f16_saturation_count(1 if reset else 0))


def volume_to_f16(vol_cl: torch.Tensor, vis_sum: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 volume [B,D,H,W,8] (divided by vis_sum + 1e-6 first when given) -> fp16 in a new buffer, clamped to the fp16 range."""
    B, D, H, W, G = vol_cl.shape
    out = torch.empty(B, D, H, W, G, dtype=torch.float16, device=vol_cl.device)
    check(lib().mvs_volume_to_f16(ptr(vol_cl), ptr(vis_sum), ptr(out), B, D, H, W, G, stream_of(vol_cl)), "mvs_volume_to_f16")
    return out


def slab_pack(vol_cl: torch.Tensor, vis_sum: torch.Tensor, send_bufs, rows) -> None:
    """One launch: send_bufs[j] (or None) <- rows[j] = [r0, r1) of the partial volume [B,D,H,W,G] followed by the same rows of the
    partial visibility sum [B,H,W] (the messages of the slab exchange)."""
    import ctypes as C
    B, D, H, W, G = vol_cl.shape
    n = len(send_bufs)
    fn = lib().mvs_slab_pack          # FIRST: fetching the entry point clears the per-call device record that ptr() fills
    ptrs = (C.c_void_p * n)(*[None if b is None else ptr(b) for b in send_bufs])
    r0 = (C.c_int * n)(*[int(r[0]) for r in rows])
    r1 = (C.c_int * n)(*[int(r[1]) for r in rows])
    check(fn(ptr(vol_cl), ptr(vis_sum), C.cast(ptrs, C.c_void_p), C.cast(r0, C.c_void_p), C.cast(r1, C.c_void_p), n, B, D, H, W, G,
             stream_of(vol_cl)), "mvs_slab_pack")


def slab_reduce(vol_cl: torch.Tensor, vis_sum: torch.Tensor, recv_bufs, my_rank: int, out: torch.Tensor, r0: int, r1: int) -> torch.Tensor:
    """One launch: out <- sum over ranks, in rank order, of their partials of rows [r0, r1): the own slice read in place, the others
    from recv_bufs[j] (None = rank j sent nothing)."""
    import ctypes as C
    B, D, H, W, G = vol_cl.shape
    n = len(recv_bufs)
    fn = lib().mvs_slab_reduce        # before the pointer array, see slab_pack
    ptrs = (C.c_void_p * n)(*[None if b is None else ptr(b) for b in recv_bufs])
    check(fn(ptr(vol_cl), ptr(vis_sum), C.cast(ptrs, C.c_void_p), n, my_rank, ptr(out), r0, r1, B, D, H, W, G, stream_of(vol_cl)),
          "mvs_slab_reduce")
    return out


def to_split(x_cl: torch.Tensor) -> torch.Tensor:
    """fp32 channel-last [..., C] -> the split activation format of MVS_PREC_BF16X3_SPLIT, same shape and dtype (the bytes are, per
    voxel, C / 8 octets of [hi x8 | lo x8] bf16 with hi = bf16(x), lo = bf16(x - hi)).  Plain torch ops: a test / tooling utility,
    the product path never converts (the kernels' epilogues write the format)."""
    C = x_cl.shape[-1]
    hi = x_cl.to(torch.bfloat16)
    lo = (x_cl - hi.float()).to(torch.bfloat16)
    pair = torch.stack([hi.reshape(*x_cl.shape[:-1], C // 8, 8), lo.reshape(*x_cl.shape[:-1], C // 8, 8)], dim=-2)    # [..., C/8, 2, 8]
    return pair.contiguous().view(torch.float32).reshape(x_cl.shape)


def from_split(s_cl: torch.Tensor) -> torch.Tensor:
    """Inverse of to_split up to the split's 2^-17 relative rounding: hi + lo as fp32."""
    C = s_cl.shape[-1]
    pair = s_cl.contiguous().view(torch.bfloat16).reshape(*s_cl.shape[:-1], C // 8, 2, 8).float()
    return (pair[..., 0, :] + pair[..., 1, :]).reshape(s_cl.shape)


# ---- a7-a9 --------------------------------------------------------------------------------------
def _act_dtype(x_cl: torch.Tensor, precision: int):
    """Element type of the activation tensors of a call: fp16 for MVS_PREC_F16X2, fp32 (plain or split pairs) otherwise."""
    want = torch.float16 if precision in _lib.F16_CODES else torch.float32
    if x_cl.dtype != want:
        raise _lib.MvsHipError("precision %d takes %s activations, got %s" % (precision, want, x_cl.dtype))
    return want


def conv3d_bn_relu(x_cl: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, kd: int,
                   stride: Tuple[int, int, int], relu: bool = True, precision: int = 0) -> torch.Tensor:
    B, D, H, W, cin = x_cl.shape
    sd, sh, sw = stride
    pd = kd // 2
    od, oh, ow = (D + 2 * pd - kd) // sd + 1, (H - 1) // sh + 1, (W - 1) // sw + 1
    y = torch.empty(B, od, oh, ow, cout, dtype=_act_dtype(x_cl, precision), device=x_cl.device)
    check(lib().mvs_conv3d_bn_relu_fwd(ptr(x_cl), ptr(w_packed), ptr(bias), ptr(y), B, cin, cout, D, H, W, kd, sd, sh, sw,
                                       1 if relu else 0, precision, stream_of(x_cl)), "mvs_conv3d_bn_relu_fwd")
    return y


def deconv3d_bn_relu_add(x_cl: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, cout: int, sd: int,
                         skip_cl: Optional[torch.Tensor] = None, precision: int = 0) -> torch.Tensor:
    B, D, H, W, cin = x_cl.shape
    y = torch.empty(B, D * sd, 2 * H, 2 * W, cout, dtype=_act_dtype(x_cl, precision), device=x_cl.device)
    if skip_cl is not None:
        assert tuple(skip_cl.shape) == tuple(y.shape) and skip_cl.dtype == y.dtype, "skip tensor shape / dtype mismatch"
    check(lib().mvs_deconv3d_bn_relu_add_fwd(ptr(x_cl), ptr(w_packed), ptr(bias), ptr(skip_cl), ptr(y), B, cin, cout, D, H, W,
                                             sd, precision, stream_of(x_cl)), "mvs_deconv3d_bn_relu_add_fwd")
    return y


def conv3d_is_tuned(cin: int, cout: int, kd: int, stride: Tuple[int, int, int]) -> bool:
    """A tuned MFMA kernel exists for Conv3d(cin, cout, (kd,3,3), stride, padding (kd//2,1,1)) (the table of csrc/conv_cfg.h)."""
    return bool(lib().mvs_conv3d_is_tuned(cin, cout, kd, *stride))


def deconv3d_is_tuned(cin: int, cout: int, sd: int) -> bool:
    return bool(lib().mvs_deconv3d_is_tuned(cin, cout, sd))


def conv3d_generic(x_cl: torch.Tensor, w_tck: torch.Tensor, bias: Optional[torch.Tensor], cout: int, ksize: Tuple[int, int, int],
                   stride: Tuple[int, int, int], padding: Tuple[int, int, int], relu: bool = False, skip_cl: Optional[torch.Tensor] = None,
                   transposed: bool = False, output_padding: Tuple[int, int, int] = (0, 0, 0)) -> torch.Tensor:
    """Shape-generic exact-fp32 Conv3d / ConvTranspose3d + bias + ReLU + skip (mvs_conv3d_generic_fwd): x_cl [B,D,H,W,Cin] fp32 ->
    [B,OD,OH,OW,cout] fp32; w_tck = packing.pack_generic_conv_weights / pack_generic_deconv_weights ([taps][Cin][Cout])."""
    if x_cl.dtype != torch.float32:
        raise _lib.MvsHipError("the generic convolution takes fp32 activations, got %s" % x_cl.dtype)
    B, D, H, W, cin = x_cl.shape
    k, s, p = tuple(ksize), tuple(stride), tuple(padding)
    if tuple(w_tck.shape) != (k[0] * k[1] * k[2], cin, cout) or w_tck.dtype != torch.float32:
        raise _lib.MvsHipError("generic convolution weights %s %s do not match fp32 [%d][%d][%d]" % (tuple(w_tck.shape), w_tck.dtype, k[0] * k[1] * k[2], cin, cout))
    if bias is not None and (bias.dtype != torch.float32 or bias.numel() < cout):
        raise _lib.MvsHipError("generic convolution bias must be fp32 with at least %d entries, got %s %s" % (cout, bias.dtype, tuple(bias.shape)))
    if transposed:
        out = [(n - 1) * s[i] - 2 * p[i] + k[i] + output_padding[i] for i, n in enumerate((D, H, W))]
    else:
        out = [(n + 2 * p[i] - k[i]) // s[i] + 1 for i, n in enumerate((D, H, W))]
    y = torch.empty(B, out[0], out[1], out[2], cout, dtype=torch.float32, device=x_cl.device)
    if skip_cl is not None and (tuple(skip_cl.shape) != tuple(y.shape) or skip_cl.dtype != torch.float32):
        raise _lib.MvsHipError("skip tensor %s / %s does not match the layer output %s fp32" % (tuple(skip_cl.shape), skip_cl.dtype, tuple(y.shape)))
    check(lib().mvs_conv3d_generic_fwd(ptr(x_cl), ptr(w_tck), ptr(bias), ptr(skip_cl), ptr(y), B, cin, cout, D, H, W, out[0], out[1], out[2],
                                       k[0], k[1], k[2], s[0], s[1], s[2], p[0], p[1], p[2], 1 if transposed else 0, 1 if relu else 0,
                                       stream_of(x_cl)), "mvs_conv3d_generic_fwd")
    return y


def deconv3d_prob(x_cl: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, sd: int, skip_cl: torch.Tensor, prob_w: torch.Tensor,
                  prob_b: torch.Tensor, precision: int) -> torch.Tensor:
    """Last U-Net layer (Cout = 8) + skip + 1x1x1 `prob` in one launch -> logits [B, D*sd, 2H, 2W]."""
    B, D, H, W, cin = x_cl.shape
    _act_dtype(x_cl, precision)
    _act_dtype(skip_cl, precision)
    logits = torch.empty(B, D * sd, 2 * H, 2 * W, dtype=torch.float32, device=x_cl.device)
    check(lib().mvs_deconv3d_prob_fwd(ptr(x_cl), ptr(w_packed), ptr(bias), ptr(skip_cl), ptr(prob_w), ptr(prob_b), ptr(logits), B, cin, D, H, W,
                                      sd, precision, stream_of(x_cl)), "mvs_deconv3d_prob_fwd")
    return logits


# ---- section 8f #2: training-mode regulariser ----------------------------------------------------
def deconv3d_linear(x_cl: torch.Tensor, w_packed: torch.Tensor, zero_bias: torch.Tensor, cout: int, sd: int, precision: int) -> torch.Tensor:
    """ConvTranspose3d(k3, padding 1, stride (sd,2,2), output_padding (sd-1,1,1)) without bias / BatchNorm / ReLU."""
    B, D, H, W, cin = x_cl.shape
    y = torch.empty(B, D * sd, 2 * H, 2 * W, cout, dtype=torch.float32, device=x_cl.device)
    check(lib().mvs_deconv3d_linear_fwd(ptr(x_cl), ptr(w_packed), ptr(zero_bias), ptr(y), B, cin, cout, D, H, W, sd, precision, stream_of(x_cl)),
          "mvs_deconv3d_linear_fwd")
    return y


def bn_stats(x_cl: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """-> float64 [groups, 2C] (or [2C]): per-channel [sum x | sum x^2] of each of `groups` equal slices along the leading axis."""
    Cc = x_cl.shape[-1]
    sums = torch.empty(groups, 2 * Cc, dtype=torch.float64, device=x_cl.device)
    check(lib().mvs_bn_stats(ptr(x_cl), ptr(sums), x_cl.numel() // Cc // groups, Cc, groups, stream_of(x_cl)), "mvs_bn_stats")
    return sums if groups > 1 else sums[0]


def bn_finalize(sums: torch.Tensor, count: float, eps: float, running_mean=None, running_var=None, momentum: float = 0.0):
    """-> (mean, biased var, invstd), each [C] or [groups, C]; with running statistics given, their momentum step(s) - one per group,
    in order - happen in the same launch."""
    groups = sums.shape[0] if sums.dim() == 2 else 1
    Cc = sums.shape[-1] // 2
    buf = torch.empty(3, groups, Cc, dtype=torch.float32, device=sums.device)
    check(lib().mvs_bn_finalize(ptr(sums), float(count), float(eps), ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(running_mean), ptr(running_var),
                                float(momentum), Cc, groups, stream_of(sums)), "mvs_bn_finalize")
    if sums.dim() == 2:
        return buf[0], buf[1], buf[2]
    return buf[0, 0], buf[1, 0], buf[2, 0]


def bn_running_update(mean, var, count: float, momentum: float, running_mean, running_var) -> None:
    groups = mean.shape[0] if mean.dim() == 2 else 1
    check(lib().mvs_bn_running_update(ptr(mean), ptr(var), float(count), float(momentum), ptr(running_mean), ptr(running_var), mean.shape[-1],
                                      groups, stream_of(mean)), "mvs_bn_running_update")


def _bn_groups(mean):
    return mean.shape[0] if mean.dim() == 2 else 1


def bn_relu_apply(z_cl, mean, invstd, gamma, beta, skip_cl=None, relu=True) -> torch.Tensor:
    """mean / invstd [C], or [groups, C] for `groups` equal slices of z_cl along its leading axis."""
    Cc, groups = z_cl.shape[-1], _bn_groups(mean)
    y = torch.empty_like(z_cl)
    check(lib().mvs_bn_relu_apply(ptr(z_cl), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(skip_cl), ptr(y), z_cl.numel() // Cc // groups, Cc,
                                  1 if relu else 0, groups, stream_of(z_cl)), "mvs_bn_relu_apply")
    return y


def bn_relu_bwd_reduce(dy_cl, z_cl, mean, invstd, gamma, beta, relu=True) -> torch.Tensor:
    """-> float64 [2C] (or [groups, 2C]): [sum g | sum g * xhat] with g = dy through the ReLU mask (= [d beta | d gamma] of these voxels)."""
    Cc, groups = z_cl.shape[-1], _bn_groups(mean)
    sums = torch.empty(groups, 2 * Cc, dtype=torch.float64, device=z_cl.device)
    check(lib().mvs_bn_relu_bwd(ptr(dy_cl), ptr(z_cl), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(sums), 1.0, None,
                                z_cl.numel() // Cc // groups, Cc, 1 if relu else 0, 1, 0, groups, stream_of(z_cl)), "mvs_bn_relu_bwd")
    return sums if mean.dim() == 2 else sums[0]


def bn_relu_bwd_apply(dy_cl, z_cl, mean, invstd, gamma, beta, sums, count: float, relu=True, use_batch_stats=True) -> torch.Tensor:
    Cc, groups = z_cl.shape[-1], _bn_groups(mean)
    dz = torch.empty_like(z_cl)
    check(lib().mvs_bn_relu_bwd(ptr(dy_cl), ptr(z_cl), ptr(mean), ptr(invstd), ptr(gamma), ptr(beta), ptr(sums), float(count), ptr(dz),
                                z_cl.numel() // Cc // groups, Cc, 1 if relu else 0, 1 if use_batch_stats else 0, 1, groups, stream_of(z_cl)),
          "mvs_bn_relu_bwd")
    return dz


class TrainWorkspace:
    """Scratch shared by the block calls of one autograd Function: zero bias, packed-weight buffer, statistics sums."""

    def __init__(self, device):
        self.zero_bias = torch.zeros(64, dtype=torch.float32, device=device)
        self.wpack = torch.empty(64 * 64 * 32 * 2, dtype=torch.bfloat16, device=device)      # >= any packed U-Net weight (64 x 64 x 27, hi + lo)
        self.sums = torch.empty(16 * 2 * 64, dtype=torch.float64, device=device)             # up to 16 statistics groups


def _out_dims(transposed, kd, stride, D, H, W):
    sd, sh, sw = stride
    if transposed:
        return D * sd, 2 * H, 2 * W
    return (D + 2 * (kd // 2) - kd) // sd + 1, (H - 1) // sh + 1, (W - 1) // sw + 1


def train_block_fwd(ws: TrainWorkspace, a_in, w, transposed: bool, kd: int, stride, gamma, beta, eps: float, running_mean, running_var,
                    momentum: float, skip=None, groups: int = 1):
    """conv / transposed conv + batch-statistics BatchNorm + ReLU [+ skip] in one C call -> (z, stats [3, groups, C], y)."""
    B, D, H, W, cin = a_in.shape
    cout = w.shape[1] if transposed else w.shape[0]
    od, oh, ow = _out_dims(transposed, kd, stride, D, H, W)
    z = torch.empty(B, od, oh, ow, cout, dtype=torch.float32, device=a_in.device)
    y = torch.empty_like(z)
    stats = torch.empty(3, groups, cout, dtype=torch.float32, device=a_in.device)
    check(lib().mvs_train_block_fwd(ptr(a_in), ptr(w), 1 if transposed else 0, cin, cout, kd, stride[0], stride[1], stride[2], B, D, H, W,
                                    ptr(gamma), ptr(beta), float(eps), ptr(running_mean), ptr(running_var), float(momentum), ptr(skip),
                                    ptr(ws.zero_bias), ptr(ws.wpack), ptr(ws.sums), ptr(z), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(y),
                                    groups, stream_of(a_in)), "mvs_train_block_fwd")
    return z, stats, y


def train_block_bwd(ws: TrainWorkspace, dy, a_in, z, stats, w, transposed: bool, kd: int, stride, gamma, beta, running_mean, running_var,
                    momentum: float, need_da: bool = True, groups: int = 1):
    """Backward of train_block_fwd in one C call -> (dw, dgamma, dbeta, da or None)."""
    B, D, H, W, cin = a_in.shape
    cout = z.shape[-1]
    dz = torch.empty_like(z)
    dw = torch.empty(w.shape[0], w.shape[1], kd, 3, 3, dtype=torch.float32, device=z.device)
    dgb = torch.empty(2, cout, dtype=torch.float32, device=z.device)
    da = torch.empty_like(a_in) if need_da else None
    check(lib().mvs_train_block_bwd(ptr(dy), ptr(a_in), ptr(z), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(w), 1 if transposed else 0, cin, cout, kd,
                                    stride[0], stride[1], stride[2], B, D, H, W, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
                                    float(momentum), ptr(ws.zero_bias), ptr(ws.wpack), ptr(ws.sums), ptr(dz), ptr(dw), ptr(dgb[0]), ptr(dgb[1]), ptr(da),
                                    groups, stream_of(z)), "mvs_train_block_bwd")
    return dw, dgb[0], dgb[1], da


def pack_conv_weights_device(w: torch.Tensor, ch: int, tflip: bool = False) -> torch.Tensor:
    """packing.pack_conv_weights_bf16x3 in one launch on the weight's device.  w [Cout, Cin, kd, 3, 3]; tflip: the packed weight is
    W'[co][ci][tap] = w[ci][co][reversed tap] (w then has shape [Cin', Cout'] = [W' columns, W' rows])."""
    w = _f32c(w)
    cout, cin = (w.shape[1], w.shape[0]) if tflip else (w.shape[0], w.shape[1])
    ntap = w.shape[2] * w.shape[3] * w.shape[4]
    n = lib().mvs_pack_conv_weights_elems(cout, cin, ntap, ch)
    if n < 0:
        raise _lib.MvsHipError("pack_conv_weights_device: unsupported shape %s / chunk %d" % (tuple(w.shape), ch))
    out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    check(lib().mvs_pack_conv_weights(ptr(w), ptr(out), cout, cin, ntap, ch, 1 if tflip else 0, stream_of(w)), "mvs_pack_conv_weights")
    return out


def pack_deconv_weights_device(w: torch.Tensor, sd: int) -> torch.Tensor:
    """packing.pack_deconv_weights_bf16x3 in one launch.  w [Cin, Cout, 3, 3, 3]."""
    w = _f32c(w)
    cin, cout = w.shape[:2]
    n = lib().mvs_pack_deconv_weights_elems(cin, cout, sd)
    if n < 0:
        raise _lib.MvsHipError("pack_deconv_weights_device: unsupported shape %s" % (tuple(w.shape),))
    out = torch.empty(n, dtype=torch.bfloat16, device=w.device)
    check(lib().mvs_pack_deconv_weights(ptr(w), ptr(out), cin, cout, sd, stream_of(w)), "mvs_pack_deconv_weights")
    return out


def conv3d_wgrad(a_cl: torch.Tensor, g_cl: torch.Tensor, stride: Tuple[int, int, int], kd: int = 3) -> torch.Tensor:
    """Weight gradient of Conv3d(k (kd,3,3), 'same' padding, stride): a_cl [B,D,H,W,CA] input, g_cl [B,OD,OH,OW,CB] output gradient
    -> [CB, CA, kd, 3, 3]."""
    B, D, H, W, CA = a_cl.shape
    CB = g_cl.shape[-1]
    sd, sh, sw = stride
    assert tuple(g_cl.shape[:4]) == (B, (D - 1) // sd + 1, (H - 1) // sh + 1, (W - 1) // sw + 1), "wgrad: gradient shape does not match the stride"
    dw = torch.empty(CB, CA, kd, 3, 3, dtype=torch.float32, device=a_cl.device)
    check(lib().mvs_conv3d_wgrad(ptr(a_cl), ptr(g_cl), ptr(dw), B, CA, CB, D, H, W, kd, sd, sh, sw, stream_of(a_cl)), "mvs_conv3d_wgrad")
    return dw


def conv3d_logits(x_cl: torch.Tensor, w_packed: torch.Tensor, bias: torch.Tensor, precision: int) -> torch.Tensor:
    """CostRegNet's 3x3x3 `prob` head on the MFMA path: x_cl [B,D,H,W,8] -> logits [B,D,H,W]."""
    B, D, H, W, c = x_cl.shape
    assert c == 8
    _act_dtype(x_cl, precision)
    logits = torch.empty(B, D, H, W, dtype=torch.float32, device=x_cl.device)
    check(lib().mvs_conv3d_logits_fwd(ptr(x_cl), ptr(w_packed), ptr(bias), ptr(logits), B, D, H, W, precision, stream_of(x_cl)),
          "mvs_conv3d_logits_fwd")
    return logits


def _ptr_array(ts: Sequence[torch.Tensor]):
    return (C.c_void_p * len(ts))(*[ptr(t) for t in ts])


def regnet(kind: int, volume_cl: torch.Tensor, w_packed: Sequence[torch.Tensor], bias: Sequence[torch.Tensor],
           precision: int = 0) -> torch.Tensor:
    """Whole U-Net up to (not including) `prob`: volume_cl [B,D,H,W,8] -> feat_cl [B,D,H,W,8]."""
    B, D, H, W, c = volume_cl.shape
    assert c == 8 and len(w_packed) == 9 and len(bias) == 9
    _act_dtype(volume_cl, precision)
    out = torch.empty_like(volume_cl)
    nbytes = lib().mvs_regnet_workspace_bytes(kind, B, D, H, W)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=volume_cl.device)
    wa, ba = _ptr_array(w_packed), _ptr_array(bias)
    check(lib().mvs_regnet_fwd(kind, ptr(volume_cl), C.cast(wa, C.c_void_p), C.cast(ba, C.c_void_p), ptr(out), ptr(ws), nbytes,
                               B, D, H, W, precision, stream_of(volume_cl)), "mvs_regnet_fwd")
    return out


def regnet_logits(kind: int, volume_cl: torch.Tensor, w_packed: Sequence[torch.Tensor], bias: Sequence[torch.Tensor],
                  prob_w: torch.Tensor, prob_b: torch.Tensor, precision: int) -> torch.Tensor:
    """U-Net + fused 1x1x1 `prob` head: volume_cl [B,D,H,W,8] -> logits [B,D,H,W] (the feature volume stays on chip)."""
    B, D, H, W, c = volume_cl.shape
    assert c == 8 and len(w_packed) == 9 and len(bias) == 9
    _act_dtype(volume_cl, precision)
    logits = torch.empty(B, D, H, W, dtype=torch.float32, device=volume_cl.device)
    nbytes = lib().mvs_regnet_workspace_bytes(kind, B, D, H, W)
    ws = torch.empty(max(nbytes // 4, 1), dtype=torch.float32, device=volume_cl.device)
    wa, ba = _ptr_array(w_packed), _ptr_array(bias)
    check(lib().mvs_regnet_logits_fwd(kind, ptr(volume_cl), C.cast(wa, C.c_void_p), C.cast(ba, C.c_void_p), ptr(prob_w), ptr(prob_b),
                                      ptr(logits), ptr(ws), nbytes, B, D, H, W, precision, stream_of(volume_cl)), "mvs_regnet_logits_fwd")
    return logits


# ---- a10/a11 ------------------------------------------------------------------------------------
def prob_regress(feat_cl: torch.Tensor, prob_w: torch.Tensor, prob_b: Optional[torch.Tensor], ksize: int, hyp: torch.Tensor,
                 tmp: float, mode: int, conf_n: int = 0, want_volumes: bool = True):
    B, D, H, W, _ = feat_cl.shape
    dev = feat_cl.device
    depth = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    conf = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    need_pre = want_volumes or D not in (4, 8, 16, 32, 48)
    pv = torch.empty(B, D, H, W, dtype=torch.float32, device=dev) if want_volumes else None
    pre = torch.empty(B, D, H, W, dtype=torch.float32, device=dev) if need_pre else None
    check(lib().mvs_prob_regress_fwd(ptr(feat_cl), ptr(prob_w), ptr(prob_b), ksize, ptr(hyp), float(tmp), mode, conf_n,
                                     ptr(depth), ptr(conf), ptr(pv), ptr(pre), B, D, H, W, stream_of(feat_cl)),
          "mvs_prob_regress_fwd")
    return depth, conf, pv, pre


def softmax_regress(logits: torch.Tensor, hyp: torch.Tensor, tmp: float, mode: int, conf_n: int = 0, want_prob: bool = True,
                    conf_prev: Optional[Sequence[torch.Tensor]] = None):
    """-> (depth, conf, prob_volume or None).  conf_prev (round 5): the EARLIER cascade stages' confidence maps (power-of-two smaller);
    the head then also writes the cascade's averaged confidence (DINOv2_mvsformer_model.py:167-177, confidence_average fused into the
    last stage's head) and returns it as a fourth value."""
    lg, hp = _f32c(logits), _f32c(hyp)
    B, D, H, W = lg.shape
    depth = torch.empty(B, H, W, dtype=torch.float32, device=lg.device)
    conf = torch.empty(B, H, W, dtype=torch.float32, device=lg.device)
    pv = torch.empty_like(lg) if want_prob else None
    if conf_prev is None:
        check(lib().mvs_softmax_regress_fwd(ptr(lg), ptr(hp), float(tmp), mode, conf_n, ptr(depth), ptr(conf), ptr(pv), B, D, H, W,
                                            stream_of(lg)), "mvs_softmax_regress_fwd")
        return depth, conf, pv
    prev = [_f32c(c) for c in conf_prev]
    shifts = []
    for c in prev:
        s = (H // c.shape[1]).bit_length() - 1
        assert c.shape[0] == B and c.shape[1] << s == H and c.shape[2] << s == W, "confidence maps must be power-of-two downsamplings"
        shifts.append(s)
    avg = torch.empty(B, H, W, dtype=torch.float32, device=lg.device)
    pa = _ptr_array(prev) if prev else None
    sa = (C.c_int * max(len(shifts), 1))(*shifts)
    check(lib().mvs_softmax_regress_confavg_fwd(ptr(lg), ptr(hp), float(tmp), mode, conf_n, ptr(depth), ptr(conf), ptr(pv),
                                                C.cast(pa, C.c_void_p) if prev else None, C.cast(sa, C.c_void_p), len(prev), ptr(avg), B, D, H, W,
                                                stream_of(lg)), "mvs_softmax_regress_confavg_fwd")
    return depth, conf, pv, avg


def softmax_regress_schedule(logits: torch.Tensor, hyp: torch.Tensor, tmp: float, mode: int, conf_n: int, want_prob: bool, next_D: int, ratio: float):
    """Round 5: softmax_regress + schedule_inverse_range(shift=False) of the NEXT stage in one launch
    -> (depth, conf, prob_volume or None, next_hyp [B,next_D,2H,2W])."""
    lg, hp = _f32c(logits), _f32c(hyp)
    B, D, H, W = lg.shape
    depth = torch.empty(B, H, W, dtype=torch.float32, device=lg.device)
    conf = torch.empty(B, H, W, dtype=torch.float32, device=lg.device)
    pv = torch.empty_like(lg) if want_prob else None
    nxt = torch.empty(B, next_D, 2 * H, 2 * W, dtype=torch.float32, device=lg.device)
    check(lib().mvs_softmax_regress_schedule_fwd(ptr(lg), ptr(hp), float(tmp), mode, conf_n, ptr(depth), ptr(conf), ptr(pv), float(ratio), ptr(nxt), next_D,
                                                 B, D, H, W, stream_of(lg)), "mvs_softmax_regress_schedule_fwd")
    return depth, conf, pv, nxt


# ---- section 8f #1: stage-1 transformer regulariser ----------------------------------------------
def position3d(K: torch.Tensor, hyp: torch.Tensor, depth_values: torch.Tensor, pe_range: Optional[torch.Tensor] = None):
    """get_position_3d(normalize=True): K [B,3,3], hyp [B,D,H,W], depth_values [B,n] -> (position3d [B,3,D,H,W],
    range [6] = height_min, height_max, width_min, width_max, depth_min, depth_max).  ``pe_range`` = a previous
    call's range tensor to reuse its first four entries (later cascade stages)."""
    Kc, hp, dv = _f32c(K), _f32c(hyp), _f32c(depth_values)
    B, D, H, W = hp.shape
    compute = pe_range is None
    rng = torch.empty(6, dtype=torch.float32, device=hp.device) if compute else pe_range.clone()
    wsb = lib().mvs_position3d_workspace_bytes()
    ws = torch.empty(wsb // 4, dtype=torch.float32, device=hp.device)
    pos = torch.empty(B, 3, D, H, W, dtype=torch.float32, device=hp.device)
    check(lib().mvs_position3d_fwd(ptr(Kc), ptr(hp), ptr(dv), dv.numel(), ptr(rng), 1 if compute else 0, ptr(ws), wsb, ptr(pos),
                                   B, D, H, W, stream_of(hp)), "mvs_position3d_fwd")
    return pos, rng


def position3d_raw(K: torch.Tensor, hyp: torch.Tensor) -> torch.Tensor:
    """get_position_3d(normalize=False): K [B,3,3], hyp [B,D,H,W] -> the frustum points [B,3,D,H,W] = K^-1 [x, y, 1] * depth."""
    Kc, hp = _f32c(K), _f32c(hyp)
    B, D, H, W = hp.shape
    pos = torch.empty(B, 3, D, H, W, dtype=torch.float32, device=hp.device)
    check(lib().mvs_position3d_raw_fwd(ptr(Kc), ptr(hp), ptr(pos), B, D, H, W, stream_of(hp)), "mvs_position3d_raw_fwd")
    return pos


def position_encoding3d(position3d: torch.Tensor, Cc: int, rescale: float = 4.0) -> torch.Tensor:
    """PositionEncoding3D as a tensor: position3d [B,3,D,H,W] -> [B,3C,D,H,W] (position_encoding.py:164-189)."""
    pos = _f32c(position3d)
    B, three, D, H, W = pos.shape
    if three != 3:
        raise _lib.MvsHipError("position3d must be [B,3,D,H,W], got %s" % (tuple(pos.shape),))
    import math
    if Cc < 2 or Cc % 2:
        raise _lib.MvsHipError("PositionEncoding3D needs an even channel count, got %d" % Cc)
    # the frequency table exactly as the reference builds it (fp32 exp of fp32 products, position_encoding.py:169)
    div = torch.exp(torch.arange(0, Cc, 2).float() * (-math.log(10000.0) / Cc)).to(pos.device)
    pe = torch.empty(B, 3 * Cc, D, H, W, dtype=torch.float32, device=pos.device)
    check(lib().mvs_position_encoding3d_fwd(ptr(pos), ptr(div), ptr(pe), B, Cc, float(rescale), D * H * W, stream_of(pos)),
          "mvs_position_encoding3d_fwd")
    return pe


def tr_embed(volume_cl: torch.Tensor, pos: Optional[torch.Tensor], pe_w, pe_div, w_packed, bias, ln_w, ln_b, rate, precision: int):
    B, D, H, W, Cc = volume_cl.shape
    assert Cc == 8
    rd, rh, rw = rate
    n = (D // rd) * (H // rh) * (W // rw)
    tokens = torch.empty(B, n, 64, dtype=torch.float32, device=volume_cl.device)
    div = (C.c_float * 4)(*[float(v) for v in pe_div]) if pos is not None else None
    check(lib().mvs_tr_embed_fwd(ptr(volume_cl), ptr(_f32c(pos)) if pos is not None else None, ptr(pe_w) if pos is not None else None,
                                 C.cast(div, C.c_void_p) if div is not None else None, ptr(w_packed), ptr(bias), ptr(ln_w), ptr(ln_b),
                                 ptr(tokens), B, D, H, W, rd, rh, rw, precision, stream_of(volume_cl)), "mvs_tr_embed_fwd")
    return tokens


def tr_linear(x: torch.Tensor, w_packed, bias, epilogue: int, N: int, precision: int, residual=None, gamma=None, ln_w=None, ln_b=None,
              ln_eps: float = 1e-5):
    B, n, K = x.shape
    y = torch.empty(B, n, N, dtype=torch.float32, device=x.device)
    check(lib().mvs_tr_linear_fwd(ptr(x), ptr(w_packed), ptr(bias), epilogue, ptr(residual), ptr(gamma), ptr(ln_w), ptr(ln_b),
                                  float(ln_eps), ptr(y), B, n, K, N, precision, stream_of(x)), "mvs_tr_linear_fwd")
    return y


def tr_attention(x: torch.Tensor, wqkv_packed, heads: int, softmax_scale: float, precision: int, attn_precision: Optional[int] = None) -> torch.Tensor:
    """x [B,n,64] -> softmax(q k^T * scale) v for all heads, [B,n,64] (qkv projection + flash attention, two launches).
    ``attn_precision`` (default = precision): PREC_ATTN16 = one fp16 term per operand like the reference's flash-attn (the module's
    default), PREC_BF16X3 = fp32-equivalent split-bf16 products, PREC_BF16P = the latter with bf16 probabilities in p.v."""
    B, n, Cc = x.shape
    ap = precision if attn_precision is None else attn_precision
    nb = lib().mvs_tr_attention_operand_bytes(B, n, heads)
    buf = torch.empty(3, nb // 2, dtype=torch.bfloat16, device=x.device)
    check(lib().mvs_tr_qkv_fwd(ptr(x), ptr(wqkv_packed), ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), float(softmax_scale), B, n, heads,
                               precision, ap, stream_of(x)), "mvs_tr_qkv_fwd")
    out = torch.empty(B, n, Cc, dtype=torch.float32, device=x.device)
    check(lib().mvs_tr_attention_fwd(ptr(buf[0]), ptr(buf[1]), ptr(buf[2]), ptr(out), B, n, heads, ap, stream_of(x)),
          "mvs_tr_attention_fwd")
    return out


def tr_attention_bwd(qkv: torch.Tensor, o: torch.Tensor, d_o: torch.Tensor, heads: int, softmax_scale: float) -> torch.Tensor:
    """Backward of softmax(scale q k^T) v: qkv [B,n,3*heads*16] (the projection's plain output), o / d_o [B,n,heads*16] -> d_qkv like qkv."""
    B, n, _ = qkv.shape
    qkv, o, d_o = _f32c(qkv), _f32c(o), _f32c(d_o)
    d_qkv = torch.empty_like(qkv)
    ws = torch.empty(2, B * heads * n, dtype=torch.float32, device=qkv.device)
    check(lib().mvs_tr_attention_bwd(ptr(qkv), ptr(o), ptr(d_o), ptr(d_qkv), ptr(ws[0]), ptr(ws[1]), B, n, heads, float(softmax_scale),
                                     stream_of(qkv)), "mvs_tr_attention_bwd")
    return d_qkv


def tr_up_prob(tokens: torch.Tensor, w_packed, up_bias, ln_w, ln_b, prob_w, prob_b, dhw, rate, precision: int) -> torch.Tensor:
    B = tokens.shape[0]
    D, H, W = dhw
    logits = torch.empty(B, D, H, W, dtype=torch.float32, device=tokens.device)
    check(lib().mvs_tr_up_prob_fwd(ptr(tokens), ptr(w_packed), ptr(up_bias), ptr(ln_w), ptr(ln_b), ptr(prob_w), ptr(prob_b), ptr(logits),
                                   B, D, H, W, rate[0], rate[1], rate[2], precision, stream_of(tokens)), "mvs_tr_up_prob_fwd")
    return logits


# ---- section 8f #3: depth-map filtering ------------------------------------------------------------
def fusion_pack_cams(cams: torch.Tensor) -> torch.Tensor:
    """cams [..., 2, 4, 4] -> packed {K, K^-1, E, E^-1} [N, campack] for the fusion kernel."""
    c = _f32c(cams).reshape(-1, 2, 4, 4)
    nf = lib().mvs_fusion_campack_floats()
    out = torch.empty(c.shape[0], nf, dtype=torch.float32, device=c.device)
    check(lib().mvs_fusion_prepare_cams(ptr(c), c.shape[0], ptr(out), stream_of(c)), "mvs_fusion_prepare_cams")
    return out


def fusion_filter(dynamic: bool, ref_depth, srcs_depth, ref_cam, srcs_cam, *, ref_conf=None, srcs_conf=None, xyd_in=None, in_range_in=None,
                  conf_thresh: float = 0.0, p0: float = 1.0, p1: float = 0.01, vthresh: float = 0.0, want_xyd: bool = False,
                  want_filter: bool = True, want_masks: bool = False, want_points: bool = True):
    """ref_depth [n,h,w]; srcs_depth [n,v,h,w] (or None with xyd_in [n,v,3,h,w]); cameras [n,2,4,4] / [n,v,2,4,4]."""
    n, h, w = ref_depth.shape
    dev = ref_depth.device
    if xyd_in is not None:
        xyd_in = _f32c(xyd_in)
        v = xyd_in.shape[1]
    else:
        v = srcs_depth.shape[1]
    rc = fusion_pack_cams(ref_cam) if ref_cam is not None else None
    sc = fusion_pack_cams(srcs_cam) if srcs_cam is not None else None
    out = {}
    if want_xyd:
        out["reproj_xyd"] = torch.empty(n, v, 3, h, w, dtype=torch.float32, device=dev)
        if not dynamic:
            out["in_range"] = torch.empty(n, v, h, w, dtype=torch.float32, device=dev)
    if want_filter:
        out["depth"] = torch.empty(n, h, w, dtype=torch.float32, device=dev)
        out["geo_mask"] = torch.empty(n, h, w, dtype=torch.uint8, device=dev)
        out["mask"] = torch.empty(n, h, w, dtype=torch.uint8, device=dev)
        if want_masks:
            out["vis_masks"] = torch.empty((n, v, v - 1, h, w) if dynamic else (n, v, h, w), dtype=torch.uint8, device=dev)
        if want_points and rc is not None:
            out["points"] = torch.empty(n, 3, h, w, dtype=torch.float32, device=dev)
    g = out.get
    check(lib().mvs_fusion_filter_fwd(1 if dynamic else 0, ptr(ref_depth), ptr(ref_conf), ptr(srcs_depth), ptr(srcs_conf), ptr(rc), ptr(sc),
                                      ptr(xyd_in), ptr(in_range_in), float(conf_thresh), float(p0), float(p1), float(vthresh),
                                      ptr(g("reproj_xyd")), ptr(g("in_range")), ptr(g("vis_masks")), ptr(g("depth")), ptr(g("geo_mask")),
                                      ptr(g("mask")), ptr(g("points")), n, v, h, w, stream_of(ref_depth)), "mvs_fusion_filter_fwd")
    return out


def fusion_ave(ref_depth, reproj_xyd, masks):
    n, h, w = ref_depth.shape
    x = _f32c(reproj_xyd)
    out = torch.empty(n, h, w, dtype=torch.float32, device=ref_depth.device)
    check(lib().mvs_fusion_ave_fwd(ptr(ref_depth), ptr(x), ptr(masks), ptr(out), n, x.shape[1], h, w, stream_of(ref_depth)), "mvs_fusion_ave_fwd")
    return out


# ---- a13-a16 ------------------------------------------------------------------------------------
def depth_regression(p: torch.Tensor, depth_values: torch.Tensor) -> torch.Tensor:
    pp, dv = _f32c(p), _f32c(depth_values)
    B, D, H, W = pp.shape
    out = torch.empty(B, H, W, dtype=torch.float32, device=pp.device)
    check(lib().mvs_depth_regression_fwd(ptr(pp), ptr(dv), ptr(out), B, D, H, W, stream_of(pp)), "mvs_depth_regression_fwd")
    return out


def conf_regression(p: torch.Tensor, n: int) -> torch.Tensor:
    pp = _f32c(p)
    B, D, H, W = pp.shape
    out = torch.empty(B, H, W, dtype=torch.float32, device=pp.device)
    check(lib().mvs_conf_regression_fwd(ptr(pp), int(n), ptr(out), B, D, H, W, stream_of(pp)), "mvs_conf_regression_fwd")
    return out


def init_range(depth_values: torch.Tensor, ndepths: int, H: int, W: int, inverse: bool) -> torch.Tensor:
    """depth_values [B,N] (one range per image) or [B,H,W,N] (one per pixel, module.py:683-688 / 698-703) -> hypotheses [B,D,H,W]."""
    dv = _f32c(depth_values)
    if dv.dim() == 4:
        B, h, w, N = dv.shape
        if (h, w) != (H, W):
            raise _lib.MvsHipError("per-pixel initial ranges %s do not match the %dx%d hypothesis maps" % (tuple(dv.shape), H, W))
        hyp = torch.empty(B, ndepths, H, W, dtype=torch.float32, device=dv.device)
        check(lib().mvs_init_range_pixel_fwd(ptr(dv), N, 1 if inverse else 0, ptr(hyp), B, ndepths, H, W, stream_of(dv)), "mvs_init_range_pixel_fwd")
        return hyp
    B, N = dv.shape
    hyp = torch.empty(B, ndepths, H, W, dtype=torch.float32, device=dv.device)
    check(lib().mvs_init_range_fwd(ptr(dv), N, 1 if inverse else 0, ptr(hyp), B, ndepths, H, W, stream_of(dv)), "mvs_init_range_fwd")
    return hyp


def schedule_inverse_range(prev_depth: torch.Tensor, prev_hyp: torch.Tensor, ndepths: int, ratio: float, H: int, W: int,
                           shift: bool = False) -> torch.Tensor:
    d, h = _f32c(prev_depth), _f32c(prev_hyp)
    B, Dp = h.shape[:2]
    assert tuple(d.shape[-2:]) == (H // 2, W // 2) and tuple(h.shape[-2:]) == (H // 2, W // 2)
    hyp = torch.empty(B, ndepths, H, W, dtype=torch.float32, device=d.device)
    check(lib().mvs_schedule_inverse_range_fwd(ptr(d), ptr(h), Dp, float(ratio), 1 if shift else 0, ptr(hyp), B, ndepths, H, W, stream_of(d)),
          "mvs_schedule_inverse_range_fwd")
    return hyp


def schedule_range(prev_depth: torch.Tensor, ndepths: int, interval: torch.Tensor, H: int, W: int) -> torch.Tensor:
    d = _f32c(prev_depth)
    B = d.shape[0]
    per_pixel = interval.dim() == 3                       # [B,H/2,W/2] intervals (module.py:731-732 takes them as they are)
    if per_pixel:
        if tuple(interval.shape) != tuple(d.shape):
            raise _lib.MvsHipError("per-pixel depth intervals %s do not match the depth map %s" % (tuple(interval.shape), tuple(d.shape)))
        itv = _f32c(interval)
    else:
        itv = _f32c(interval.reshape(-1).expand(B) if interval.numel() == 1 else interval.reshape(B))
    hyp = torch.empty(B, ndepths, H, W, dtype=torch.float32, device=d.device)
    check(lib().mvs_schedule_range_fwd(ptr(d), ptr(itv), 1 if per_pixel else 0, ptr(hyp), B, ndepths, H, W, stream_of(d)), "mvs_schedule_range_fwd")
    return hyp


def confidence_average(confs: Sequence[torch.Tensor], H: int, W: int) -> torch.Tensor:
    confs = [_f32c(c) for c in confs]
    B = confs[0].shape[0]
    shifts = []
    for c in confs:
        s = (H // c.shape[1]).bit_length() - 1
        assert c.shape[1] << s == H and c.shape[2] << s == W, "confidence maps must be power-of-two downsamplings"
        shifts.append(s)
    out = torch.empty(B, H, W, dtype=torch.float32, device=confs[0].device)
    pa = _ptr_array(confs)
    sa = (C.c_int * len(shifts))(*shifts)
    check(lib().mvs_confidence_average(C.cast(pa, C.c_void_p), C.cast(sa, C.c_void_p), len(confs), ptr(out), B, H, W,
                                       stream_of(out)), "mvs_confidence_average")
    return out


def ncdhw_to_cl(x: torch.Tensor) -> torch.Tensor:
    x = _f32c(x)
    B, Cc, D, H, W = x.shape
    y = torch.empty(B, D, H, W, Cc, dtype=torch.float32, device=x.device)
    check(lib().mvs_ncdhw_to_cl(ptr(x), ptr(y), B, Cc, D, H, W, stream_of(x)), "mvs_ncdhw_to_cl")
    return y


def cl_to_ncdhw(x_cl: torch.Tensor) -> torch.Tensor:
    B, D, H, W, Cc = x_cl.shape
    y = torch.empty(B, Cc, D, H, W, dtype=torch.float32, device=x_cl.device)
    check(lib().mvs_cl_to_ncdhw(ptr(x_cl), ptr(y), B, Cc, D, H, W, stream_of(x_cl)), "mvs_cl_to_ncdhw")
    return y
