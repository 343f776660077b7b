"""Mirror of the reference's depth-map filtering step: ``misc/fusion.py`` (the reprojection
consistency filters) and the per-view body of the two drivers in ``test.py`` (``filter_depth`` :388-409, ``dynamic_filter_depth``
:455-483), on one fused HIP kernel (csrc/fusion_kernels.hip).  Same function names, argument meaning and tensor shapes as the
reference:

    get_reproj(ref_depth [n,1,h,w], srcs_depth [n,v,1,h,w], ref_cam [n,2,4,4], srcs_cam [n,v,2,4,4]) -> (reproj_xyd [n,v,3,h,w], in_range [n,v,1,h,w])
    vis_filter(ref_depth, reproj_xyd, in_range, img_dist_thresh, depth_thresh, vthresh)               -> (masks [n,v,1,h,w], mask [n,1,h,w])
    ave_fusion(ref_depth, reproj_xyd, masks)                                                            -> [n,1,h,w]
    get_reproj_dynamic(...) -> reproj_xyd;  vis_filter_dynamic(ref_depth, reproj_xyd, dist_base, rel_diff_base) -> (masks [n,v,v-1,h,w], mask [n,v,1,h,w])

``filter_depth`` / ``dynamic_filter_depth`` run a whole reference view in ONE launch without materialising the reprojection.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def _maps(ref_depth, srcs_depth):
    n, v, _, h, w = srcs_depth.shape
    return ops._f32c(ref_depth).reshape(n, h, w), ops._f32c(srcs_depth).reshape(n, v, h, w), (n, v, h, w)


def get_reproj(ref_depth, srcs_depth, ref_cam, srcs_cam):
    rd, sd, (n, v, h, w) = _maps(ref_depth, srcs_depth)
    out = ops.fusion_filter(False, rd, sd, ref_cam, srcs_cam, want_xyd=True, want_filter=False)
    return out["reproj_xyd"], out["in_range"].reshape(n, v, 1, h, w)


def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):
    rd, sd, _ = _maps(ref_depth, srcs_depth)
    return ops.fusion_filter(True, rd, sd, ref_cam, srcs_cam, want_xyd=True, want_filter=False)["reproj_xyd"]


def vis_filter(ref_depth, reproj_xyd, in_range, img_dist_thresh, depth_thresh, vthresh):
    n, v, _, h, w = reproj_xyd.shape
    out = ops.fusion_filter(False, ops._f32c(ref_depth).reshape(n, h, w), None, None, None, xyd_in=reproj_xyd,
                            in_range_in=ops._f32c(in_range).reshape(n, v, h, w), p0=img_dist_thresh, p1=depth_thresh, vthresh=vthresh,
                            want_masks=True, want_points=False)
    return out["vis_masks"].reshape(n, v, 1, h, w).to(ref_depth.dtype), out["geo_mask"].reshape(n, 1, h, w).bool()


def vis_filter_dynamic(ref_depth, reproj_xyd, dist_base=4, rel_diff_base=1300):
    n, v, _, h, w = reproj_xyd.shape
    out = ops.fusion_filter(True, ops._f32c(ref_depth).reshape(n, h, w), None, None, None, xyd_in=reproj_xyd, p0=dist_base, p1=rel_diff_base,
                            want_masks=True, want_points=False)
    masks = out["vis_masks"].reshape(n, v, v - 1, h, w).bool()
    return masks, masks[:, :, -1:]


def ave_fusion(ref_depth, reproj_xyd, masks):
    n, v, _, h, w = reproj_xyd.shape
    return ops.fusion_ave(ops._f32c(ref_depth).reshape(n, h, w), reproj_xyd, ops._f32c(masks).reshape(n, v, h, w)).reshape(n, 1, h, w)


def filter_depth(ref_depth, ref_conf, srcs_depth, srcs_conf, ref_cam, srcs_cam, *, conf_thresh, thres_disp, thres_view,
                 depth_thresh=0.01) -> Dict[str, torch.Tensor]:
    """One reference view of test.py:388-409: source depths gated by their confidence, static thresholds, averaged depth,
    final mask and world points.  Maps as in the reference's loader: ref_depth [n,1,h,w], ref_conf [n,h,w], srcs_depth
    [n,v,1,h,w], srcs_conf [n,v,h,w]."""
    rd, sd, (n, v, h, w) = _maps(ref_depth, srcs_depth)
    out = ops.fusion_filter(False, rd, sd, ref_cam, srcs_cam, ref_conf=ops._f32c(ref_conf).reshape(n, h, w),
                            srcs_conf=None if srcs_conf is None else ops._f32c(srcs_conf).reshape(n, v, h, w), conf_thresh=conf_thresh,
                            p0=thres_disp, p1=depth_thresh, vthresh=thres_view)
    return {"depth": out["depth"].reshape(n, 1, h, w), "geo_mask": out["geo_mask"].reshape(n, 1, h, w).bool(),
            "mask": out["mask"].reshape(n, 1, h, w).bool(), "points": out["points"]}


def dynamic_filter_depth(ref_depth, ref_conf, srcs_depth, ref_cam, srcs_cam, *, conf_thresh, dist_base=4, rel_diff_base=1300) -> Dict[str, torch.Tensor]:
    """One reference view of test.py:455-483 (dynamic consistency checking)."""
    rd, sd, (n, v, h, w) = _maps(ref_depth, srcs_depth)
    out = ops.fusion_filter(True, rd, sd, ref_cam, srcs_cam, ref_conf=ops._f32c(ref_conf).reshape(n, h, w), conf_thresh=conf_thresh,
                            p0=dist_base, p1=rel_diff_base)
    return {"depth": out["depth"].reshape(n, 1, h, w), "geo_mask": out["geo_mask"].reshape(n, 1, h, w).bool(),
            "mask": out["mask"].reshape(n, 1, h, w).bool(), "points": out["points"]}
