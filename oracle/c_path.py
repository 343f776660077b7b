"""ORACLE (test infrastructure, NOT the product): one cascade stage composed from the plain-C operator restatements
in oracle/stage_ref.c with numpy.  Mirrors ``oracle/ref_path.stage_forward`` (and through it the reference's
``StageNet.forward``, cost_volume.py:51-133) without calling a single torch operator, so the two oracles check each
other: torch's ATen kernels vs the operators' published definitions.  Slow (direct loops): small shapes only.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
_fp = C.POINTER(C.c_float)


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "libstage_ref.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        _LIB = C.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def compose_proj(pm):                       # cost_volume.py:68-71
    P = pm[0].copy()
    P[:3, :4] = pm[1][:3, :3] @ pm[0][:3, :4]
    return P


def homography(src_pm, ref_pm):             # warping.py:80-82, fp32 like the reference
    M = (compose_proj(src_pm) @ np.linalg.inv(compose_proj(ref_pm))).astype(np.float32)
    return _f32(np.concatenate([M[:3, :3].reshape(9), M[:3, 3]]))


def homo_warp(src, hom, depth):
    src, hom, depth = _f32(src), _f32(hom), _f32(depth)
    Cc, H, W = src.shape
    D = depth.shape[0]
    out = np.empty((Cc, D, H, W), np.float32)
    mask = np.empty((D, H, W), np.uint8)
    lib().ref_homo_warp(_p(src), _p(hom), _p(depth), _p(out), mask.ctypes.data_as(C.POINTER(C.c_uint8)), Cc, D, H, W)
    return out, mask.astype(bool)


def _bn_relu(y, bn, relu):
    n = int(np.prod(y.shape[1:]))
    lib().ref_bn_relu(_p(y), _p(bn["weight"]), _p(bn["bias"]), _p(bn["running_mean"]), _p(bn["running_var"]), y.shape[0],
                      C.c_size_t(n), 1 if relu else 0)
    return y


def conv_bn_relu_3d(x, w, bn, stride, relu=True):
    x = _f32(x)
    ci, D, H, W = x.shape
    co = w.shape[0]
    sd, sh, sw = stride
    od, oh, ow = (D - 1) // sd + 1, (H - 1) // sh + 1, (W - 1) // sw + 1
    y = np.empty((co, od, oh, ow), np.float32)
    lib().ref_conv3d(_p(x), _p(w), None, _p(y), ci, co, D, H, W, 3, sd, sh, sw)
    return _bn_relu(y, bn, relu)


def deconv_bn_relu_3d(x, w, bn, stride):
    x = _f32(x)
    ci, D, H, W = x.shape
    co = w.shape[1]
    sd, sh, sw = stride
    y = np.empty((co, D * sd, H * sh, W * sw), np.float32)
    lib().ref_conv_transpose3d(_p(x), _p(w), _p(y), ci, co, D, H, W, sd, sh, sw)
    return _bn_relu(y, bn, True)


def _bn(sd, prefix):
    return {k: _f32(sd[prefix + "." + k]) for k in ("weight", "bias", "running_mean", "running_var")}


def regulariser(vol, sd, prefix="cost_reg"):
    """CostRegNet (module.py:398-408) or CostRegNet3D (module.py:494-504), chosen by the state-dict keys."""
    p = prefix + "."
    three_d = (p + "conv7.0.weight") in sd
    s = (1, 2, 2) if three_d else (2, 2, 2)
    one = (1, 1, 1)

    def conv(x, name, stride):
        return conv_bn_relu_3d(x, _f32(sd[p + name + ".conv.weight"]), _bn(sd, p + name + ".bn"), stride)

    def deconv(x, name):
        if three_d:
            return deconv_bn_relu_3d(x, _f32(sd[p + name + ".0.weight"]), _bn(sd, p + name + ".1"), s)
        return deconv_bn_relu_3d(x, _f32(sd[p + name + ".conv.weight"]), _bn(sd, p + name + ".bn"), s)

    conv0 = _f32(vol)
    conv2 = conv(conv(conv0, "conv1", s), "conv2", one)
    conv4 = conv(conv(conv2, "conv3", s), "conv4", one)
    x = conv(conv(conv4, "conv5", s), "conv6", one)
    x = conv4 + deconv(x, "conv7")
    x = conv2 + deconv(x, "conv9")
    x = _f32(conv0 + deconv(x, "conv11"))
    ci, D, H, W = x.shape
    y = np.empty((1, D, H, W), np.float32)
    pw = _f32(sd[p + "prob.weight"])
    if three_d:                                                                   # 1x1x1 + bias, module.py:486
        y[0] = np.tensordot(pw.reshape(ci), x, axes=(0, 0)) + np.float32(np.asarray(sd[p + "prob.bias"]).reshape(-1)[0])
    else:                                                                         # 3x3x3 no bias, module.py:391
        lib().ref_conv3d(_p(x), _p(pw), None, _p(y), ci, 1, D, H, W, 3, 1, 1, 1)
    return y[0]


def vis_weight(entropy, sd, prefix="vis"):
    x = _f32(entropy)[None]
    H, W = x.shape[-2:]
    for i in range(3):
        w = _f32(sd["%s.%d.conv.weight" % (prefix, i)])
        y = np.empty((w.shape[0], H, W), np.float32)
        lib().ref_conv2d(_p(_f32(x)), _p(w), None, _p(y), w.shape[1], w.shape[0], H, W, 3)
        x = _bn_relu(y, _bn(sd, "%s.%d.bn" % (prefix, i)), True)
    w4, b4 = _f32(sd[prefix + ".3.weight"]).reshape(-1), _f32(sd[prefix + ".3.bias"])
    z = np.tensordot(w4, x, axes=(0, 0)) + b4[0]
    return (1.0 / (1.0 + np.exp(-z))).astype(np.float32)


def stage_forward(features, proj, hyp, tmp, sd, G=8):
    """features [V,C,H,W], proj [V,2,4,4], hyp [D,H,W] (batch 1) -> dict of numpy arrays."""
    sd = {k: (v.numpy() if hasattr(v, "numpy") else np.asarray(v)) for k, v in sd.items()}
    feats, proj, hyp = _f32(features), _f32(proj), _f32(hyp)
    V, Cc, H, W = feats.shape
    D = hyp.shape[0]
    vol_sum = np.zeros((G, D, H, W), np.float32)
    vis_sum = np.zeros((H, W), np.float32)
    for v in range(1, V):
        hom = homography(proj[v], proj[0])
        warped, _ = homo_warp(feats[v], hom, hyp)
        ip = np.empty((G, D, H, W), np.float32)
        lib().ref_group_corr(_p(_f32(feats[0])), _p(warped), _p(ip), Cc, G, D, H, W)
        ent = np.empty((H, W), np.float32)
        lib().ref_entropy(_p(ip), _p(ent), G, D, H * W)
        w = vis_weight(ent, sd)
        vol_sum = vol_sum + ip * w[None, None]
        vis_sum = vis_sum + w
    vol = (vol_sum / (vis_sum[None, None] + np.float32(1e-6))).astype(np.float32)
    logit = _f32(regulariser(vol, sd))
    prob = np.empty((D, H, W), np.float32)
    depth = np.empty((H, W), np.float32)
    conf = np.empty((H, W), np.float32)
    lib().ref_softmax_regress(_p(logit), _p(hyp), C.c_float(tmp), _p(prob), _p(depth), _p(conf), D, H * W)
    return {"volume_mean": vol, "prob_volume_pre": logit, "prob_volume": prob, "depth": depth, "photometric_confidence": conf}
