"""Mirror of the part of the reference's ``models/position_encoding.py`` that the depth hot path uses: the frustum coordinates
(``get_position_3d``, position_encoding.py:138-161, normalised or not) and the Frustoconical position encoding built from them
(``PositionEncoding3D``, :164-189).  On the hot path the encoding is evaluated inside the patch-embedding kernel
(csrc/transformer_kernels.hip) and never materialised; the function here is the standalone form with the reference's call shape."""
from __future__ import annotations

import torch

from . import ops


def get_position_3d(B, H, W, K, depth_values, depth_min, depth_max, height_min, height_max, width_min, width_max, normalize=True):
    """Same call form and return value as the reference: K [B,3,3], depth_values [B,D,H,W] (the stage's hypotheses),
    depth_min / depth_max scalars or 0-dim tensors, the four range values None (measure them) or a previous call's.
    -> (position3d [B,3,D,H,W], height_min, height_max, width_min, width_max)."""
    if not normalize:                      # position_encoding.py:150: the points as they are, the range arguments handed back untouched
        return ops.position3d_raw(K, depth_values), height_min, height_max, width_min, width_max
    dev = depth_values.device
    lim = torch.stack([torch.as_tensor(depth_min, dtype=torch.float32, device=dev).reshape(()),
                       torch.as_tensor(depth_max, dtype=torch.float32, device=dev).reshape(())])
    given = [height_min, height_max, width_min, width_max]
    rng = None
    if not any(v is None for v in given):                                    # position_encoding.py:152 measures all four or none
        rng = torch.stack([torch.as_tensor(v, dtype=torch.float32, device=dev).reshape(()) for v in given] + [lim[0], lim[1]])
    pos, rng = ops.position3d(K, depth_values, lim, rng)
    return pos, rng[0], rng[1], rng[2], rng[3]


def PositionEncoding3D(position3d, C, rescale=4.0):
    """position3d [B,3,D,H,W] in 0..1 -> [B,3C,D,H,W]: per axis C channels of sin / cos pairs at the frequencies
    exp(2f * (-ln 10000 / C)) scaled by `rescale` (reference position_encoding.py:164-189).  One HIP kernel; forward form."""
    if torch.is_grad_enabled() and position3d.requires_grad:
        raise NotImplementedError("PositionEncoding3D is a forward form (the reference builds position3d under no_grad, position_encoding.py:140)")
    return ops.position_encoding3d(position3d, int(C), float(rescale))
