"""Host-side parameter preparation for the HIP kernels: BatchNorm folding and MFMA weight packing.

Packed conv layout (read by ``conv3d_mfma_kernel``; DESIGN.md "MFMA weight packing"):

    K-quads are enumerated per channel chunk ("pass", CH input channels) tap-major:
        kq = tap * (CH/4) + cq,   tap = (kd*3 + kh)*3 + kw,   cq = channel quad inside the chunk
    and consumed four at a time ("step"); lane group g = lane >> 4 of the wave owns quad 4*step + g.
    packed[pass][step][mb][lane = g*16 + j][s] = W'[cout = 16*mb + j][cin = pass*CH + 4*cq + s][tap]
    (zero where the quad index or cout runs past the end).  W' = W * gamma / sqrt(var + eps).

Packed deconv layout (``deconv3d_mfma_kernel``): all 27 taps, 16-channel blocks q:
    packed[tap][q][mb][lane = g*16 + j][s] = Wt'[cin = 16*q + 4*g + s][cout = 16*mb + j][tap]
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch

BN_EPS = 1e-5


def fold_bn(weight: torch.Tensor, bn: Dict[str, torch.Tensor], out_dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """Fold eval-mode BatchNorm into the preceding bias-free conv.  ``out_dim`` = axis of ``weight`` that is Cout."""
    scale = bn["weight"].double() / torch.sqrt(bn["running_var"].double() + float(bn.get("eps", BN_EPS)))
    shift = bn["bias"].double() - bn["running_mean"].double() * scale
    shape = [1] * weight.dim()
    shape[out_dim] = -1
    return (weight.double() * scale.reshape(shape)).float(), shift.float()


def conv_chunk(cin: int, stride: Tuple[int, int, int]) -> int:
    """Channel chunk staged per LDS pass - must match the ConvCfg table in csrc/conv_kernels.hip."""
    return 16 if tuple(stride) == (1, 1, 1) and cin >= 16 else 8


def pack_conv_weights(w: torch.Tensor, ch: int) -> torch.Tensor:
    """w [Cout, Cin, kd, 3, 3] (BN already folded) -> packed fp32 1-D tensor."""
    cout, cin = w.shape[:2]
    ntap = w.shape[2] * w.shape[3] * w.shape[4]
    assert cin % ch == 0 and ch % 4 == 0
    qc = ch // 4
    npass = cin // ch
    nquad = ntap * qc
    nstep = (nquad + 3) // 4
    mrep = (cout + 15) // 16
    wt = w.reshape(cout, npass, qc, 4, ntap).float()                       # [co, pass, cq, s, tap]
    wt = wt.permute(1, 4, 2, 0, 3).reshape(npass, nquad, cout, 4)          # [pass, kq = tap*qc + cq, co, s]
    full = torch.zeros(npass, nstep * 4, mrep * 16, 4, dtype=torch.float32, device=w.device)
    full[:, :nquad, :cout] = wt
    full = full.reshape(npass, nstep, 4, mrep, 16, 4)                      # [pass, step, g, mb, j, s]
    return full.permute(0, 1, 3, 2, 4, 5).contiguous().reshape(-1)         # [pass, step, mb, g, j, s]


def pack_deconv_weights(w: torch.Tensor) -> torch.Tensor:
    """w [Cin, Cout, 3, 3, 3] (ConvTranspose3d layout, BN folded over Cout) -> packed fp32 1-D tensor."""
    cin, cout = w.shape[:2]
    assert cin % 16 == 0 and tuple(w.shape[2:]) == (3, 3, 3)
    nq = cin // 16
    mrep = (cout + 15) // 16
    wt = w.reshape(nq, 4, 4, cout, 27).float()                             # [q, g, s, co, tap]
    full = torch.zeros(nq, 4, 4, mrep * 16, 27, dtype=torch.float32, device=w.device)
    full[:, :, :, :cout] = wt
    full = full.reshape(nq, 4, 4, mrep, 16, 27)                            # [q, g, s, mb, j, tap]
    return full.permute(5, 0, 3, 1, 4, 2).contiguous().reshape(-1)         # [tap, q, mb, g, j, s]


def _split_bf16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [...] -> bf16 [2, ...] with x ~= hi + lo (both round-to-nearest-even), the operand format of the bf16x3 kernels."""
    hi = x.to(torch.bfloat16)
    lo = (x - hi.float()).to(torch.bfloat16)
    return torch.stack([hi, lo])


def _split_f16(x: torch.Tensor) -> torch.Tensor:
    """fp32 [...] -> fp16 [2, ...] with x ~= hi + lo (22 significant bits; lo flushes below 6e-8): operands of the f16x2 kernels,
    viewed as bf16 so that the packers below stay dtype-agnostic (2-byte elements, bit pattern preserved)."""
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    return torch.stack([hi, lo]).view(torch.bfloat16)


def f16x2(packer, *args):
    """One of the pack_*_bf16x3 functions with the fp16 hi + lo split instead of the bf16 one (the weights of MVS_PREC_F16X2)."""
    return packer(*args, split=_split_f16)


def pack_conv_weights_bf16x3(w: torch.Tensor, ch: int, split=None) -> torch.Tensor:
    """w [Cout, Cin, kd, 3, 3] (BN folded) -> bf16 1-D tensor for ``conv3d_mfma_bf16x3_kernel``:

        packed[pass][step][mb][hi|lo][lane = g*16 + j][e] = W'[16*mb + j][pass*CH + 8*oc + e][tap],
        (tap, oc) = divmod(4*step + g, CH/8)      (channel octets enumerated tap-major, four per step)
    """
    cout, cin = w.shape[:2]
    ntap = w.shape[2] * w.shape[3] * w.shape[4]
    assert cin % ch == 0 and ch % 8 == 0
    opt = ch // 8
    npass = cin // ch
    noct = ntap * opt
    nstep = (noct + 3) // 4
    mrep = (cout + 15) // 16
    wt = w.reshape(cout, npass, opt, 8, ntap).float().permute(1, 4, 2, 0, 3).reshape(npass, noct, cout, 8)   # [pass, o, co, e]
    full = torch.zeros(npass, nstep * 4, mrep * 16, 8, dtype=torch.float32, device=w.device)
    full[:, :noct, :cout] = wt
    full = full.reshape(npass, nstep, 4, mrep, 16, 8).permute(0, 1, 3, 2, 4, 5)                              # [pass, step, mb, g, j, e]
    return (split or _split_bf16)(full).permute(1, 2, 3, 0, 4, 5, 6).contiguous().reshape(-1)                           # [pass, step, mb, 2, g, j, e]


def pack_linear_bf16x3(w: torch.Tensor, split=None) -> torch.Tensor:
    """w [N, K] (y = x @ w.T; N % 16 == 0, K % 32 == 0) -> bf16 1-D tensor for ``tr_gemm_kernel``:

        packed[step][mb][hi|lo][lane = g*16 + j][e] = w[16*mb + j][32*step + 8*g + e]
    """
    n, k = w.shape
    assert n % 16 == 0 and k % 32 == 0, (n, k)
    full = w.float().reshape(n // 16, 16, k // 32, 4, 8).permute(2, 0, 3, 1, 4)                                # [step, mb, g, j, e]
    return (split or _split_bf16)(full).permute(1, 2, 0, 3, 4, 5).contiguous().reshape(-1)                                # [step, mb, 2, g, j, e]


def patch_embed_matrix(w: torch.Tensor) -> torch.Tensor:
    """Conv3d(kernel = stride) weight [Cout, Cin, rd, rh, rw] -> [Cout, patch_voxel*Cin + ci] (k order of the patch gather)."""
    co, ci = w.shape[:2]
    return w.float().permute(0, 2, 3, 4, 1).reshape(co, -1).contiguous()


def patch_expand_matrix(w: torch.Tensor) -> torch.Tensor:
    """ConvTranspose3d(kernel = stride) weight [Cin, Cout, rd, rh, rw] -> [patch_voxel*Cout + co, Cin]."""
    ci, co = w.shape[:2]
    return w.float().permute(2, 3, 4, 1, 0).reshape(-1, ci).contiguous()


def deconv_class_taps(sd: int):
    """Parity classes of ConvTranspose3d(k3, stride (sd,2,2)) in the kernels' order, each a list of tap indices
    (kd*3 + kh)*3 + kw ordered (a_d, a_h, a_w) with a_w fastest - mirrors ``bf_deconv_load_step``."""
    out = []
    for cls in range((2 if sd == 2 else 1) * 4):
        pw, ph, pd = cls & 1, (cls >> 1) & 1, (cls >> 2) if sd == 2 else 0
        kds = ([0, 2] if pd else [1]) if sd == 2 else [0, 1, 2]
        khs = [0, 2] if ph else [1]
        kws = [0, 2] if pw else [1]
        out.append([(kd * 3 + kh) * 3 + kw for kd in kds for kh in khs for kw in kws])
    return out


def pack_deconv_weights_bf16x3(w: torch.Tensor, sd: int, split=None) -> torch.Tensor:
    """w [Cin, Cout, 3, 3, 3] (BN folded over Cout) -> bf16 1-D tensor for ``deconv3d_mfma_bf16x3_kernel``: per parity
    class, packed[step][mb][hi|lo][lane = g*16 + j][e] = Wt'[8*oc + e][16*mb + j][taps[ti]], (ti, oc) = divmod(4*step + g, Cin/8)."""
    cin, cout = w.shape[:2]
    assert cin % 8 == 0 and tuple(w.shape[2:]) == (3, 3, 3)
    opt = cin // 8
    mrep = (cout + 15) // 16
    wf = w.reshape(opt, 8, cout, 27).float()                                       # [oc, e, co, tap]
    chunks = []
    if cout == 8:
        # Cout = 8 fills only half of the 16 MFMA rows: the two x-parity classes of a (pd, ph) pair share their input voxels
        # (pw = 1 reads inputs mx+1 and mx with kw = 0 and 2, pw = 0 reads input mx with kw = 1), so rows 0-7 carry pw = 0
        # (zero weights on the mx+1 tap) and rows 8-15 pw = 1: 2 tap units per pair instead of 1 + 2, all 16 rows useful,
        # and one lane group pair stores the 64 contiguous bytes of the two output voxels 2mx, 2mx+1.
        classes = deconv_class_taps(sd)
        for c2 in range(len(classes) // 2):
            taps1 = classes[2 * c2 + 1]                                            # pw = 1: (a_d, a_h, a_w) with a_w fastest, kw in (0, 2)
            noct = len(taps1) * opt
            nst = (noct + 3) // 4
            full = torch.zeros(nst * 4, 16, 8, dtype=torch.float32, device=w.device)
            for ti, t1 in enumerate(taps1):
                kw = t1 % 3
                full[ti * opt:(ti + 1) * opt, 8:16] = wf[:, :, :, t1].permute(0, 2, 1)           # [oc, co, e]
                if kw == 2:                                                                       # input mx: pw = 0 uses kw = 1 of the same (kd, kh)
                    full[ti * opt:(ti + 1) * opt, 0:8] = wf[:, :, :, t1 - 1].permute(0, 2, 1)
            full = full.reshape(nst, 4, 1, 16, 8).permute(0, 2, 1, 3, 4)
            chunks.append((split or _split_bf16)(full).permute(1, 2, 0, 3, 4, 5).contiguous().reshape(-1))
        return torch.cat(chunks)
    for taps in deconv_class_taps(sd):
        noct = len(taps) * opt
        nst = (noct + 3) // 4
        sel = wf[:, :, :, taps].permute(3, 0, 2, 1).reshape(noct, cout, 8)         # [o = ti*opt + oc, co, e]
        full = torch.zeros(nst * 4, mrep * 16, 8, dtype=torch.float32, device=w.device)
        full[:noct, :cout] = sel
        full = full.reshape(nst, 4, mrep, 16, 8).permute(0, 2, 1, 3, 4)             # [step, mb, g, j, e]
        chunks.append((split or _split_bf16)(full).permute(1, 2, 0, 3, 4, 5).contiguous().reshape(-1))   # [step, mb, 2, g, j, e]
    return torch.cat(chunks)


def pad_bias(b: torch.Tensor) -> torch.Tensor:
    n = max(16, ((b.numel() + 15) // 16) * 16)
    out = torch.zeros(n, dtype=torch.float32, device=b.device)
    out[: b.numel()] = b.float()
    return out


def pack_generic_conv_weights(w: torch.Tensor) -> torch.Tensor:
    """nn.Conv3d weight [Cout, Cin, kd, kh, kw] (BN already folded) -> [kd*kh*kw][Cin][Cout] fp32 (mvs_conv3d_generic_fwd)."""
    cout, cin = w.shape[:2]
    return w.float().permute(2, 3, 4, 1, 0).reshape(-1, cin, cout).contiguous()


def pack_generic_deconv_weights(w: torch.Tensor) -> torch.Tensor:
    """nn.ConvTranspose3d weight [Cin, Cout, kd, kh, kw] (BN folded along dim 1) -> [kd*kh*kw][Cin][Cout] fp32, taps NOT flipped
    (the kernel gathers input i = (o + p - k) / s for kernel index k)."""
    cin, cout = w.shape[:2]
    return w.float().permute(2, 3, 4, 0, 1).reshape(-1, cin, cout).contiguous()
