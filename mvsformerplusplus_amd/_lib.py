"""ctypes binding of libmvs_hip.so (C ABI: include/mvs_hip.h).

There is NO fallback: if the gfx950 library has not been built (``python -m mvsformerplusplus_amd.build``)
importing an op raises, and every op refuses tensors that are not on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVS_HIP_LIB: measurement scripts point the binding at a variant build of the SAME C ABI (build.build(out=...)); unset = the product library
LIB_PATH = os.environ.get("MVS_HIP_LIB") or os.path.join(_HERE, "csrc", "libmvs_hip.so")

OK = 0
ABI_VERSION = 11
TR_EPI_BIAS, TR_EPI_GELU, TR_EPI_RES_LN = 0, 1, 2
DTYPE_CODE = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}
HEAD_CE_EVAL, HEAD_CE_TRAIN, HEAD_REG = 0, 1, 2
LAYOUT_PLANAR, LAYOUT_OCTET_TILED = 0, 1
REG_COSTREGNET, REG_COSTREGNET3D = 0, 1
PREC_FP32, PREC_BF16X3, PREC_BF16P, PREC_BF16X3_SPLIT, PREC_F16X2, PREC_ATTN16, PREC_F16, PREC_F16MIX = 0, 1, 2, 3, 4, 5, 6, 7
PRECISIONS = {"fp32": PREC_FP32, "bf16x3": PREC_BF16X3, "f16x2": PREC_F16X2, "f16": PREC_F16, "f16mix": PREC_F16MIX}
F16_FORMATS = ("f16x2", "f16", "f16mix")        # conv_precision values whose regulariser ACTIVATIONS are fp16 tensors (they share packed weights)
F16_CODES = (PREC_F16X2, PREC_F16, PREC_F16MIX)
VOLUME_F32, VOLUME_SPLIT, VOLUME_F16 = 0, 1, 2
CORR_F16, CORR_F32 = 0, 1

_vp, _i, _f, _sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t

# name -> (restype, argtypes); mirrors include/mvs_hip.h one to one
SIGNATURES = {
    "mvs_abi_version": (_i, []),
    "mvs_f16_saturation_count": (C.c_ulonglong, [_i]),
    "mvs_last_error": (C.c_char_p, []),
    "mvs_compose_homography": (_i, [_vp, _i, _i, _vp, _vp]),
    "mvs_cascade_prologue_fwd": (_i, [_vp, _i, _i, _i, _vp, _vp, _i, _i, _vp, _i, _i, _i, _vp]),
    "mvs_homography_from_proj": (_i, [_vp, _vp, _i, _vp, _vp]),
    "mvs_homo_warp_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvs_warp_corr_entropy_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp] + [_i] * 10 + [_vp]),
    "mvs_pack_features": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvs_feature_conv_is_built": (_i, [_i, _i]),
    "mvs_conv2d3x3_tiles_fwd": (_i, [_vp, _i, _vp, _vp, _i, _vp, _i, _i, _i, _i, _i, _i, C.c_longlong, C.c_longlong, _vp]),
    "mvs_vis_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "mvs_vis_weight_fwd": (_i, [_vp] * 10 + [_vp, _sz, _i, _i, _i, _i, _vp]),
    "mvs_vis_conv1_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mvs_vis_out_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _vp]),
    "mvs_warp_corr_aggregate_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _i] + [_i] * 9 + [_vp]),
    "mvs_volume_normalise": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mvs_volume_to_f16": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvs_gather_is_lds_staged": (_i, [_i, _i, _i, _i, _i, _i]),
    "mvs_gather_keeps_correlations": (_i, [_i, _i, _i, _i, _i, _i]),
    "mvs_warp_corr_entropy_keep_fwd": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "mvs_corr_aggregate_fwd": (_i, [_vp, _i, _vp, _vp] + [_i] * 6 + [_vp]),
    "mvs_slab_pack": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "mvs_slab_reduce": (_i, [_vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i, _vp]),
    "mvs_conv3d_bn_relu_fwd": (_i, [_vp, _vp, _vp, _vp] + [_i] * 12 + [_vp]),
    "mvs_deconv3d_bn_relu_add_fwd": (_i, [_vp, _vp, _vp, _vp, _vp] + [_i] * 8 + [_vp]),
    "mvs_conv3d_generic_fwd": (_i, [_vp] * 5 + [_i] * 20 + [_vp]),
    "mvs_conv3d_is_tuned": (_i, [_i] * 6),
    "mvs_deconv3d_is_tuned": (_i, [_i] * 3),
    "mvs_deconv3d_prob_fwd": (_i, [_vp] * 7 + [_i] * 7 + [_vp]),
    "mvs_conv3d_logits_fwd": (_i, [_vp] * 4 + [_i] * 5 + [_vp]),
    "mvs_deconv3d_linear_fwd": (_i, [_vp] * 4 + [_i] * 8 + [_vp]),
    "mvs_bn_stats": (_i, [_vp, _vp, C.c_longlong, _i, _i, _vp]),
    "mvs_bn_finalize": (_i, [_vp, C.c_double, C.c_float, _vp, _vp, _vp, _vp, _vp, C.c_float, _i, _i, _vp]),
    "mvs_bn_running_update": (_i, [_vp, _vp, C.c_double, C.c_float, _vp, _vp, _i, _i, _vp]),
    "mvs_bn_relu_apply": (_i, [_vp] * 7 + [C.c_longlong, _i, _i, _i, _vp]),
    "mvs_bn_relu_bwd": (_i, [_vp] * 7 + [C.c_double, _vp, C.c_longlong, _i, _i, _i, _i, _i, _vp]),
    "mvs_train_block_fwd": (_i, [_vp, _vp] + [_i] * 11 + [_vp, _vp, C.c_float, _vp, _vp, C.c_float] + [_vp] * 9 + [_i, _vp]),
    "mvs_train_block_bwd": (_i, [_vp] * 7 + [_i] * 11 + [_vp, _vp, _vp, _vp, C.c_float] + [_vp] * 8 + [_i, _vp]),
    "mvs_pack_conv_weights_elems": (C.c_longlong, [_i] * 4),
    "mvs_pack_conv_weights": (_i, [_vp, _vp] + [_i] * 5 + [_vp]),
    "mvs_pack_deconv_weights_elems": (C.c_longlong, [_i] * 3),
    "mvs_pack_deconv_weights": (_i, [_vp, _vp] + [_i] * 3 + [_vp]),
    "mvs_conv3d_wgrad": (_i, [_vp] * 3 + [_i] * 10 + [_vp]),
    "mvs_warp_corr_aggregate_bwd": (_i, [_vp, _i] + [_vp] * 8 + [_i] * 7 + [_vp]),
    "mvs_regnet_workspace_bytes": (_sz, [_i, _i, _i, _i, _i]),
    "mvs_regnet_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "mvs_regnet_logits_fwd": (_i, [_i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i, _i, _i, _i, _i, _vp]),
    "mvs_prob_regress_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvs_softmax_regress_fwd": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvs_softmax_regress_schedule_fwd": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvs_softmax_regress_confavg_fwd": (_i, [_vp, _vp, _f, _i, _i, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "mvs_depth_regression_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvs_conf_regression_fwd": (_i, [_vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "mvs_init_range_fwd": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "mvs_init_range_pixel_fwd": (_i, [_vp, _i, _i, _vp, _i, _i, _i, _i, _vp]),
    "mvs_schedule_inverse_range_fwd": (_i, [_vp, _vp, _i, _f, _i, _vp, _i, _i, _i, _i, _vp]),
    "mvs_schedule_range_fwd": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _i, _vp]),
    "mvs_confidence_average": (_i, [_vp, _vp, _i, _vp, _i, _i, _i, _vp]),
    "mvs_position3d_workspace_bytes": (_sz, []),
    "mvs_position3d_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _i, _vp, _sz, _vp, _i, _i, _i, _i, _vp]),
    "mvs_position3d_raw_fwd": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvs_position_encoding3d_fwd": (_i, [_vp, _vp, _vp, _i, _i, _f, C.c_longlong, _vp]),
    "mvs_tr_embed_fwd": (_i, [_vp] * 9 + [_i] * 8 + [_vp]),
    "mvs_tr_linear_fwd": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _f, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvs_tr_attention_operand_bytes": (_sz, [_i, _i, _i]),
    "mvs_tr_qkv_fwd": (_i, [_vp, _vp, _vp, _vp, _vp, _f, _i, _i, _i, _i, _i, _vp]),
    "mvs_tr_attention_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvs_tr_attention_bwd": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp]),
    "mvs_tr_up_prob_fwd": (_i, [_vp] * 8 + [_i] * 8 + [_vp]),
    "mvs_fusion_campack_floats": (_sz, []),
    "mvs_fusion_prepare_cams": (_i, [_vp, _i, _vp, _vp]),
    "mvs_fusion_filter_fwd": (_i, [_i] + [_vp] * 8 + [_f] * 4 + [_vp] * 7 + [_i] * 4 + [_vp]),
    "mvs_fusion_ave_fwd": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp]),
    "mvs_ncdhw_to_cl": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "mvs_cl_to_ncdhw": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
}


class MvsHipError(RuntimeError):
    pass


def bind(path: str) -> C.CDLL:
    """Load a build of the C ABI and attach prototypes for every symbol the header declares."""
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.mvs_abi_version() != ABI_VERSION:
        raise MvsHipError("libmvs_hip ABI version mismatch: %d" % lib.mvs_abi_version())
    return _GuardedLib(lib)


_LIB: Optional[C.CDLL] = None
_REQUIRE_DEVICE = True        # product behaviour; only tests/hipemu flips this for host-emulated kernels


def lib() -> C.CDLL:
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise MvsHipError("%s not found: build the gfx950 HIP library first (python -m mvsformerplusplus_amd.build). "
                              "There is no CPU/PyTorch fallback for this path." % LIB_PATH)
        _LIB = bind(LIB_PATH)
    return _LIB


def check(rc: int, what: str = "") -> None:
    if rc != OK:
        msg = lib().mvs_last_error()
        raise MvsHipError("%s failed (code %d): %s" % (what or "libmvs_hip call", rc, msg.decode() if msg else "?"))


class _CallDevices(threading.local):
    """Devices of the tensors handed to ptr() while the arguments of ONE C-ABI call are evaluated - per thread (DataParallel
    replicas and autograd's per-device worker threads issue calls concurrently)."""

    def __init__(self):
        self.devs = []


_CALL_DEVICE = _CallDevices()


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """Raw address of a dense tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise MvsHipError("internal: non-contiguous tensor handed to the C ABI")
    if _REQUIRE_DEVICE and not t.is_cuda:
        raise MvsHipError("libmvs_hip needs tensors on a ROCm device (got %s); there is no CPU path" % t.device)
    if t.is_cuda:
        _CALL_DEVICE.devs.append(t.device)
    return t.data_ptr()


class _GuardedLib:
    """Every C-ABI call runs with the tensors' device made current (HIP resolves the null stream and launches on the CURRENT
    device, not on the device that owns the pointers), and refuses tensors spread over several devices - what torch's own ops
    do with their device guard.  Python evaluates `lib.mvs_x` (this __getattr__: the record is cleared, so a ptr() that raised in an
    earlier, abandoned argument list leaves nothing behind), then the arguments (ptr() records the devices), then makes the call."""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if not name.startswith("mvs_") or name in ("mvs_abi_version", "mvs_last_error") or name.endswith("_bytes") or name.endswith("_floats"):
            return fn

        del _CALL_DEVICE.devs[:]

        def call(*args):
            devs = set(_CALL_DEVICE.devs)
            del _CALL_DEVICE.devs[:]
            if len(devs) > 1:
                raise MvsHipError("%s: tensors on different devices %s" % (name, sorted(str(d) for d in devs)))
            if devs:
                dev = devs.pop()
                if torch.cuda.current_device() != dev.index:
                    with torch.cuda.device(dev):
                        return fn(*args)
            return fn(*args)
        return call


def stream_of(t: torch.Tensor) -> Optional[int]:
    """The caller's current HIP stream ON THE TENSOR'S DEVICE (work is enqueued there, like any torch op)."""
    if t.is_cuda:
        return torch.cuda.current_stream(t.device).cuda_stream
    return None
