"""Synthetic multi-view scenes for tests, golden fixtures and ``bench.py``.

No DTU / Tanks&Temples data or checkpoints exist in the build or GPU containers, so
every workload is synthetic.  The layout mirrors what the reference
dataset emits (``datasets/general_eval.py:211-242``):

* ``proj_matrices["stageK"]``: ``[B, V, 2, 4, 4]`` fp32, ``[:, :, 0]`` = extrinsic 4x4,
  ``[:, :, 1, :3, :3]`` = intrinsic 3x3; per-stage intrinsics scaled by 0.5 / 1 / 2 / 4
  relative to the quarter-resolution intrinsics.
* ``depth_values``: ``[B, numdepth]`` (only first/last element are used by the cascade).
* ``features["stageK"]``: ``[B, V, C, H/8.., W/8..]`` with C = 64/32/16/8.

Features are an analytic multi-channel texture painted on a smooth depth surface and
observed from every camera, so the group-wise correlation really peaks at the true
depth (a flat-noise volume would make depth regression degenerate).

Everything here is plain torch on the CPU; callers move tensors to the GPU.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch

STAGE_CH = (64, 32, 16, 8)          # config/mvsformer++.json feat_chs reversed per stage
STAGE_DOWN = (8, 4, 2, 1)           # stage s runs at 1/8, 1/4, 1/2, 1/1 resolution


def make_cameras(V: int, H: int, W: int, *, baseline: float = 25.0, rot_deg: float = 0.0,
                 seed: int = 0, batch: int = 1) -> torch.Tensor:
    """Full-resolution DTU-like pinhole cameras -> ``[B, V, 2, 4, 4]``.

    fx = fy = 2892.33 * (W / 1600) (DTU focal length scaled to the image width),
    principal point at the image centre, view v translated by ``baseline * v`` mm along x
    (alternating sign) and a little along y, optionally rotated by up to ``rot_deg`` degrees.
    """
    g = torch.Generator().manual_seed(seed)
    out = torch.zeros(batch, V, 2, 4, 4, dtype=torch.float32)
    f = 2892.33 * (W / 1600.0)
    for b in range(batch):
        for v in range(V):
            K = torch.eye(4, dtype=torch.float64)
            K[0, 0] = f
            K[1, 1] = f
            K[0, 2] = W / 2.0
            K[1, 2] = H / 2.0
            E = torch.eye(4, dtype=torch.float64)
            if v > 0:
                sign = 1.0 if (v % 2) else -1.0
                E[0, 3] = sign * baseline * ((v + 1) // 2)
                E[1, 3] = 0.15 * baseline * (((v * 7) % 5) - 2) / 2.0
                if rot_deg > 0:
                    ang = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 2 * math.radians(rot_deg)
                    cx, cy, cz = torch.cos(ang)
                    sx, sy, sz = torch.sin(ang)
                    Rx = torch.tensor([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=torch.float64)
                    Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)
                    Rz = torch.tensor([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=torch.float64)
                    E[:3, :3] = Rz @ Ry @ Rx
            out[b, v, 0] = E.float()
            out[b, v, 1, :3, :3] = K[:3, :3].float()
    return out


def stage_proj_matrices(full_res_proj: torch.Tensor, n_stages: int = 4) -> Dict[str, torch.Tensor]:
    """Scale the intrinsics per stage the way ``general_eval.py:229-242`` does.

    ``full_res_proj`` holds FULL-resolution intrinsics; stage s (1-based) runs at
    1/2**(n_stages - s) of full resolution, so rows 0-1 of K are divided accordingly.
    """
    out = {}
    for s in range(n_stages):
        scale = 1.0 / (2 ** (n_stages - 1 - s))
        p = full_res_proj.clone()
        p[:, :, 1, :2, :] = p[:, :, 1, :2, :] * scale
        out["stage%d" % (s + 1)] = p
    return out


def _surface_depth(xn: torch.Tensor, yn: torch.Tensor, dmin: float, dmax: float) -> torch.Tensor:
    """Smooth reference-frame depth map in [dmin, dmax] as a function of normalised pixel coords."""
    mid = 0.5 * (dmin + dmax)
    amp = 0.30 * (dmax - dmin)
    return mid + amp * (0.6 * torch.sin(2.1 * xn + 0.3) * torch.cos(1.7 * yn - 0.2) + 0.4 * torch.sin(3.3 * yn + 1.1))


def _texture(X: torch.Tensor, Y: torch.Tensor, C: int, wavelength: float, seed: int) -> torch.Tensor:
    """Analytic C-channel texture f_c(X, Y) on the world plane (mm units)."""
    g = torch.Generator().manual_seed(1000 + seed)
    k = 2 * math.pi / wavelength
    a = (torch.rand(C, 3, generator=g, dtype=torch.float64) * 2 - 1) * k
    b = (torch.rand(C, 3, generator=g, dtype=torch.float64) * 2 - 1) * k
    ph = torch.rand(C, 3, generator=g, dtype=torch.float64) * 2 * math.pi
    amp = (1.0, 0.6, 0.35)
    a, b, ph = a.to(X.device), b.to(X.device), ph.to(X.device)
    out = torch.zeros((C,) + tuple(X.shape), dtype=torch.float64, device=X.device)
    for j in range(3):
        mult = float(2 ** j)
        out += amp[j] * torch.sin(mult * (a[:, j, None, None] * X + b[:, j, None, None] * Y) + ph[:, j, None, None])
    return out


def make_features(proj_stage: torch.Tensor, C: int, H: int, W: int, *, dmin: float, dmax: float,
                  noise: float = 0.05, seed: int = 0, dtype=torch.float32, device=None) -> torch.Tensor:
    """Geometrically consistent features ``[B, V, C, H, W]`` for one stage.

    The scene is the depth surface ``_surface_depth`` seen from the reference camera.  For a
    source view the surface point behind each source pixel is found by two fixed-point
    iterations (exact for pure translations with equal z), which is accurate enough for a
    benchmark texture; a little white noise decorrelates the views.
    """
    B, V = proj_stage.shape[:2]
    device = torch.device("cpu") if device is None else torch.device(device)
    on_cpu = device.type == "cpu"
    g = torch.Generator(device=device).manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=device), torch.arange(W, dtype=torch.float64, device=device), indexing="ij")
    feats = torch.zeros(B, V, C, H, W, dtype=torch.float32, device=device)
    proj_stage = proj_stage.to(device)
    for b in range(B):
        K0 = proj_stage[b, 0, 1, :3, :3].double()
        E0 = proj_stage[b, 0, 0].double()
        # texture wavelength ~ 6 pixels of this stage at mid depth
        wl = 6.0 * (0.5 * (dmin + dmax)) / float(K0[0, 0])
        for v in range(V):
            K = proj_stage[b, v, 1, :3, :3].double()
            E = proj_stage[b, v, 0].double()
            # relative pose view v -> ref
            T = (E0.cpu() @ torch.inverse(E.cpu())).to(device)          # X_ref = T @ X_v
            rx = (xs - K[0, 2]) / K[0, 0]
            ry = (ys - K[1, 2]) / K[1, 1]
            z = torch.full_like(xs, 0.5 * (dmin + dmax))
            for _ in range(3):
                Xv = torch.stack([rx * z, ry * z, z, torch.ones_like(z)], 0).reshape(4, -1)
                Xr = (T @ Xv).reshape(4, H, W)
                ur = (Xr[0] / Xr[2]) * K0[0, 0] + K0[0, 2]
                vr = (Xr[1] / Xr[2]) * K0[1, 1] + K0[1, 2]
                zr = _surface_depth(ur / W * 2 - 1, vr / H * 2 - 1, dmin, dmax)
                # move the guess so that the ref-frame depth matches the surface
                z = z + (zr - Xr[2])
            Xv = torch.stack([rx * z, ry * z, z, torch.ones_like(z)], 0).reshape(4, -1)
            Xr = (T @ Xv).reshape(4, H, W)
            Xw = (torch.inverse(E0.cpu()).to(device) @ Xr.reshape(4, -1)).reshape(4, H, W)
            tex = _texture(Xw[0], Xw[1], C, wl, seed)
            tex = tex + noise * torch.randn(tex.shape, generator=g, dtype=torch.float64, device=device)
            feats[b, v] = tex.float()
    return feats.to(dtype)


def true_depth(proj_stage: torch.Tensor, H: int, W: int, dmin: float, dmax: float) -> torch.Tensor:
    """Ground-truth reference-view depth ``[H, W]`` of the synthetic surface."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64), torch.arange(W, dtype=torch.float64), indexing="ij")
    return _surface_depth(xs / W * 2 - 1, ys / H * 2 - 1, dmin, dmax).float()


def make_cascade_inputs(H: int, W: int, V: int, *, numdepth: int = 192, depth_min: float = 425.0,
                        depth_interval: float = 2.65, baseline: float = 25.0, rot_deg: float = 0.0,
                        seed: int = 0, batch: int = 1, feat_dtype=torch.float32,
                        stage_ch: Tuple[int, ...] = STAGE_CH, device=None) -> Tuple[Dict[str, torch.Tensor], Dict[str, torch.Tensor], torch.Tensor]:
    """(features, proj_matrices, depth_values) for a 4-stage cascade at full resolution HxW.

    ``depth_values`` mirrors ``general_eval.py:223``: arange(dmin, interval*(nd-0.5)+dmin, interval).
    H and W must be divisible by 64 (SURVEY.md §7 "odd sizes").
    """
    assert H % 64 == 0 and W % 64 == 0, "image size must be divisible by 64"
    dv = torch.arange(depth_min, depth_interval * (numdepth - 0.5) + depth_min, depth_interval, dtype=torch.float32)
    depth_values = dv[None].repeat(batch, 1)
    dmax = float(dv[-1])
    cams = make_cameras(V, H, W, baseline=baseline, rot_deg=rot_deg, seed=seed, batch=batch)
    projs = stage_proj_matrices(cams, len(stage_ch))
    feats = {}
    # keep the surface away from the ends of the hypothesis range
    lo = depth_min + 0.15 * (dmax - depth_min)
    hi = dmax - 0.15 * (dmax - depth_min)
    for s, C in enumerate(stage_ch):
        down = 2 ** (len(stage_ch) - 1 - s)
        feats["stage%d" % (s + 1)] = make_features(projs["stage%d" % (s + 1)], C, H // down, W // down,
                                                   dmin=lo, dmax=hi, seed=seed + s, dtype=feat_dtype, device=device)
    if device is not None:
        projs = {k: v.to(device) for k, v in projs.items()}
        depth_values = depth_values.to(device)
    return feats, projs, depth_values


def randomize_bn_(module: torch.nn.Module, seed: int = 0) -> None:
    """Randomise BatchNorm affine + running stats so BN folding is exercised."""
    g = torch.Generator().manual_seed(seed)
    for m in module.modules():
        if isinstance(m, (torch.nn.BatchNorm2d, torch.nn.BatchNorm3d)):
            with torch.no_grad():
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.2)


def seeded_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int) -> Dict[str, torch.Tensor]:
    """Deterministic weights for a state-dict manifest ``{key: shape}``.

    Uses numpy's frozen legacy ``RandomState`` stream with a per-key sub-seed (crc32 of the key), so
    the same key/shape/seed gives bit-identical fp32 values on every machine and numpy version and
    independent of key order.  Golden fixtures therefore store only the manifest + seed, not ~1.2 MB
    of incompressible weights per StageNet.  Conv weights ~ N(0, 2/fan_in); BN weight ~ U(.5,1.5),
    running_var ~ U(.5,1.5), running_mean / biases ~ N(0,.2).
    """
    import zlib

    import numpy as np
    out: Dict[str, torch.Tensor] = {}
    for k, shp in shapes.items():
        shp = tuple(int(s) for s in shp)
        rs = np.random.RandomState((zlib.crc32(k.encode()) + 7919 * int(seed)) % (2 ** 32))
        if k.endswith("num_batches_tracked"):
            out[k] = torch.zeros(shp, dtype=torch.int64)
            continue
        if k.endswith("running_var") or (len(shp) == 1 and k.endswith("weight")) or k.endswith(("gamma1", "gamma2")):
            a = rs.uniform(0.5, 1.5, size=shp)
        elif k.endswith("running_mean") or k.endswith("bias"):
            a = rs.standard_normal(size=shp) * 0.2
        else:
            fan_in = 1
            for s in shp[1:]:
                fan_in *= s
            a = rs.standard_normal(size=shp) * math.sqrt(2.0 / max(fan_in, 1))
        out[k] = torch.from_numpy(a.astype(np.float32))
    return out


def state_dict_manifest(sd) -> Dict[str, Tuple[int, ...]]:
    return {k: tuple(v.shape) for k, v in sd.items()}
