"""CPU restatement (torch, fp32) of the reference's depth-map filtering step - TEST INFRASTRUCTURE,
never imported by the product path.  Follows misc/fusion.py (Vis-MVSNet's filters as vendored by the reference) and the
two drivers in test.py op for op; every function cites the lines it restates.  Device-agnostic (the reference hard-codes
``.cuda()`` in get_pixel_grids, fusion.py:8-13)."""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F


def get_pixel_grids(height: int, width: int) -> torch.Tensor:                       # fusion.py:8-13   -> [h,w,3,1], pixel centres
    x = (torch.arange(width, dtype=torch.float32) + 0.5).repeat(height, 1)
    y = (torch.arange(height, dtype=torch.float32) + 0.5).repeat(width, 1).t()
    return torch.stack([x, y, torch.ones_like(x)], dim=-1).unsqueeze(-1)


def bin_op_reduce(lst: List, func):                                                 # fusion.py:16-20
    r = lst[0]
    for t in lst[1:]:
        r = func(r, t)
    return r


def idx_img2cam(idx_img_homo, depth, cam):                                           # fusion.py:23-28
    idx_cam = cam[:, 1:2, :3, :3].unsqueeze(1).inverse() @ idx_img_homo
    idx_cam = idx_cam / (idx_cam[..., -1:, :] + 1e-9) * depth.permute(0, 2, 3, 1).unsqueeze(4)
    return torch.cat([idx_cam, torch.ones_like(idx_cam[..., -1:, :])], dim=-2)


def idx_cam2world(idx_cam_homo, cam):                                                # fusion.py:31-34
    w = cam[:, 0:1, ...].unsqueeze(1).inverse() @ idx_cam_homo
    return w / (w[..., -1:, :] + 1e-9)


def idx_world2cam(idx_world_homo, cam):                                              # fusion.py:37-40
    c = cam[:, 0:1, ...].unsqueeze(1) @ idx_world_homo
    return c / (c[..., -1:, :] + 1e-9)


def idx_cam2img(idx_cam_homo, cam):                                                  # fusion.py:43-47
    idx_cam = idx_cam_homo[..., :3, :] / (idx_cam_homo[..., 3:4, :] + 1e-9)
    img = cam[:, 1:2, :3, :3].unsqueeze(1) @ idx_cam
    return img / (img[..., -1:, :] + 1e-9)


def project_img(src_img, dst_depth, src_cam, dst_cam):                               # fusion.py:50-66
    height, width = src_img.shape[-2:]
    dst_idx_img = get_pixel_grids(height, width).unsqueeze(0)
    dst2src = idx_cam2img(idx_world2cam(idx_cam2world(idx_img2cam(dst_idx_img, dst_depth, dst_cam), dst_cam), src_cam), src_cam)
    warp = dst2src[..., :2, 0].clone()
    warp[..., 0] /= width
    warp[..., 1] /= height
    warp = (warp * 2 - 1).clamp(-1.1, 1.1)
    in_range = bin_op_reduce([-1 <= warp[..., 0], warp[..., 0] <= 1, -1 <= warp[..., 1], warp[..., 1] <= 1], torch.min).to(src_img.dtype).unsqueeze(1)
    return F.grid_sample(src_img, warp, mode="bilinear", padding_mode="zeros", align_corners=True), in_range


def get_reproj(ref_depth, srcs_depth, ref_cam, srcs_cam):                            # fusion.py:80-97   n1hw, nv1hw -> nv3hw, nv1hw
    n, v, _, h, w = srcs_depth.shape
    sd = srcs_depth.reshape(n * v, 1, h, w)
    sc = srcs_cam.reshape(n * v, 2, 4, 4)
    rd = ref_depth.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 1, h, w)
    rc = ref_cam.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 2, 4, 4)
    idx_img = get_pixel_grids(h, w).unsqueeze(0)
    s2r_cam = idx_world2cam(idx_cam2world(idx_img2cam(idx_img, sd, sc), sc), rc)
    s2r_img = idx_cam2img(s2r_cam, rc)
    s2r_xyd = torch.cat([s2r_img[..., :2, 0], s2r_cam[..., 2:3, 0]], dim=-1).permute(0, 3, 1, 2)
    xyd, in_range = project_img(s2r_xyd, rd, sc, rc)
    return xyd.reshape(n, v, 3, h, w), in_range.reshape(n, v, 1, h, w)


def vis_filter(ref_depth, reproj_xyd, in_range, img_dist_thresh, depth_thresh, vthresh):      # fusion.py:100-109
    n, v, _, h, w = reproj_xyd.shape
    xy = get_pixel_grids(h, w).permute(3, 2, 0, 1).unsqueeze(1)[:, :, :2]
    dist_masks = (reproj_xyd[:, :, :2] - xy).norm(dim=2, keepdim=True) < img_dist_thresh
    depth_masks = (ref_depth.unsqueeze(1) - reproj_xyd[:, :, 2:]).abs() < (torch.max(ref_depth.unsqueeze(1), reproj_xyd[:, :, 2:]) * depth_thresh)
    masks = bin_op_reduce([in_range, dist_masks.to(ref_depth.dtype), depth_masks.to(ref_depth.dtype)], torch.min)
    mask = masks.sum(dim=1) >= (vthresh - 1.1)
    return masks, mask


def ave_fusion(ref_depth, reproj_xyd, masks):                                         # fusion.py:112-114
    return ((reproj_xyd[:, :, 2:] * masks).sum(dim=1) + ref_depth) / (masks.sum(dim=1) + 1)


def get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam):                     # fusion.py:116-153
    n, v, _, h, w = srcs_depth.shape
    sd = srcs_depth.reshape(n * v, 1, h, w)
    sc = srcs_cam.reshape(n * v, 2, 4, 4)
    rc = ref_cam.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 2, 4, 4)
    rd = ref_depth.unsqueeze(1).repeat(1, v, 1, 1, 1).reshape(n * v, 1, h, w)
    idx_img = get_pixel_grids(h, w).unsqueeze(0)
    r2s_img = idx_cam2img(idx_world2cam(idx_cam2world(idx_img2cam(idx_img, rd, rc), rc), sc), sc)
    warp = r2s_img[..., :2, 0]
    grid = torch.stack((warp[..., 0] / ((w - 1) / 2) - 1, warp[..., 1] / ((h - 1) / 2) - 1), dim=-1)
    warped = F.grid_sample(sd, grid, mode="bilinear", padding_mode="zeros", align_corners=True)
    homo = torch.cat([warp, torch.ones_like(warp[..., -1:])], dim=-1).unsqueeze(-1)
    s2r_cam = idx_world2cam(idx_cam2world(idx_img2cam(homo, warped, sc), sc), rc)
    reproj_depth = s2r_cam[:, :, :, 2, 0].clone()
    s2r_img = idx_cam2img(s2r_cam, rc)
    xyd = torch.cat([s2r_img[..., :2, 0], reproj_depth.unsqueeze(-1)], dim=-1).permute(0, 3, 1, 2)
    return xyd.reshape(n, v, 3, h, w)


def vis_filter_dynamic(ref_depth, reproj_xyd, dist_base=4, rel_diff_base=1300):       # fusion.py:156-168
    n, v, _, h, w = reproj_xyd.shape
    xy = get_pixel_grids(h, w).permute(3, 2, 0, 1).unsqueeze(1)[:, :, :2]
    corrd_diff = (reproj_xyd[:, :, :2] - xy).norm(dim=2, keepdim=True)
    depth_diff = (ref_depth.unsqueeze(1) - reproj_xyd[:, :, 2:]).abs() / ref_depth.unsqueeze(1)
    dist_thred = torch.arange(2, v + 1).reshape(1, 1, -1, 1, 1).repeat(n, v, 1, 1, 1) / dist_base
    rel_thred = torch.arange(2, v + 1).reshape(1, 1, -1, 1, 1).repeat(n, v, 1, 1, 1) / rel_diff_base
    masks = torch.min(corrd_diff < dist_thred, depth_diff < rel_thred)
    return masks, masks[:, :, -1:]


def backproject(depth, cam):                                                          # test.py:407-409 / 481-483 -> [n,3,h,w] world points
    idx_img = get_pixel_grids(*depth.shape[-2:]).unsqueeze(0)
    return idx_cam2world(idx_img2cam(idx_img, depth, cam), cam)[..., :3, 0].permute(0, 3, 1, 2)


def filter_depth(ref_depth, ref_conf, srcs_depth, srcs_conf, ref_cam, srcs_cam, *, conf_thresh, thres_disp, thres_view,
                 depth_thresh=0.01) -> Dict[str, torch.Tensor]:
    """Static filter of one reference view, test.py:388-409 ("pcd")."""
    srcs_depth = srcs_depth.clone()
    for i in range(srcs_depth.shape[1]):
        srcs_depth[:, i] *= (srcs_conf[:, i] > conf_thresh).float().unsqueeze(1)
    prob_mask = ref_conf > conf_thresh
    xyd, in_range = get_reproj(ref_depth, srcs_depth, ref_cam, srcs_cam)
    vis_masks, vis_mask = vis_filter(ref_depth, xyd, in_range, thres_disp, depth_thresh, thres_view)
    ave = ave_fusion(ref_depth, xyd, vis_masks)
    mask = bin_op_reduce([prob_mask.reshape(vis_mask.shape), vis_mask], torch.min)
    return {"reproj_xyd": xyd, "in_range": in_range, "vis_masks": vis_masks, "geo_mask": vis_mask, "depth": ave, "mask": mask,
            "points": backproject(ave, ref_cam)}


def dynamic_filter_depth(ref_depth, ref_conf, srcs_depth, ref_cam, srcs_cam, *, conf_thresh, dist_base=4, rel_diff_base=1300) -> Dict[str, torch.Tensor]:
    """Dynamic-consistency filter of one reference view, test.py:455-483 ("dpcd")."""
    v = srcs_depth.shape[1]
    dy_range = v + 1
    prob_mask = ref_conf > conf_thresh
    xyd = get_reproj_dynamic(ref_depth, srcs_depth, ref_cam, srcs_cam)
    vis_masks, vis_mask = vis_filter_dynamic(ref_depth, xyd, dist_base, rel_diff_base)
    reproj_depth = xyd[:, :, -1].clone()
    reproj_depth[~vis_mask.squeeze(2)] = 0
    geo_mask_sums = vis_masks.sum(dim=1)
    geo_mask_sum = vis_mask.sum(dim=1)
    ave = (torch.sum(reproj_depth, dim=1, keepdim=True) + ref_depth) / (geo_mask_sum + 1)
    geo_mask = geo_mask_sum >= dy_range
    for i in range(2, dy_range):
        geo_mask = torch.logical_or(geo_mask, geo_mask_sums[:, i - 2] >= i)
    mask = bin_op_reduce([prob_mask.reshape(geo_mask.shape), geo_mask], torch.min)
    return {"reproj_xyd": xyd, "vis_masks": vis_masks, "geo_mask": geo_mask, "depth": ave, "mask": mask, "points": backproject(ave, ref_cam)}
