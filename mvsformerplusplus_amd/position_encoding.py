"""Mirror of the part of the reference's ``models/position_encoding.py`` that the depth hot path uses: the normalised
frustum coordinates that feed the Frustoconical position encoding of the stage-1 transformer regulariser
(position_encoding.py:138-163).  ``PositionEncoding3D`` itself (:166-189) is evaluated inside the patch-embedding
kernel (csrc/transformer_kernels.hip) and is never materialised."""
from __future__ import annotations

import torch

from . import ops


def get_position_3d(B, H, W, K, depth_values, depth_min, depth_max, height_min, height_max, width_min, width_max, normalize=True):
    """Same call form and return value as the reference: K [B,3,3], depth_values [B,D,H,W] (the stage's hypotheses),
    depth_min / depth_max scalars or 0-dim tensors, the four range values None (measure them) or a previous call's.
    -> (position3d [B,3,D,H,W], height_min, height_max, width_min, width_max)."""
    if not normalize:
        raise NotImplementedError("get_position_3d(normalize=False) is never used by the reference's driver")
    dev = depth_values.device
    lim = torch.stack([torch.as_tensor(depth_min, dtype=torch.float32, device=dev).reshape(()),
                       torch.as_tensor(depth_max, dtype=torch.float32, device=dev).reshape(())])
    given = [height_min, height_max, width_min, width_max]
    rng = None
    if not any(v is None for v in given):                                    # position_encoding.py:152 measures all four or none
        rng = torch.stack([torch.as_tensor(v, dtype=torch.float32, device=dev).reshape(()) for v in given] + [lim[0], lim[1]])
    pos, rng = ops.position3d(K, depth_values, lim, rng)
    return pos, rng[0], rng[1], rng[2], rng[3]
