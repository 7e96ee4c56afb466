"""Ray tables on the device (reference ``src/data.py:48-73`` ray generation).

The reference builds rays on the CPU and feeds them through a ``DataLoader`` one ray at a time; at
>= 1e9 samples/s that loader is the bottleneck, so rays live in HBM as flat [M,3] tables and batches
are drawn by device-side random indices.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import torch


@dataclass
class Intrinsics:
    fx: float
    fy: float
    cx: float
    cy: float
    w: int
    h: int


def generate_rays(cameras: torch.Tensor, K: Intrinsics) -> Tuple[torch.Tensor, torch.Tensor]:
    """cameras [n,4,4] (camera-to-world) -> rays_o, rays_d of shape [n, h, w, 3].

    Pixel centres at +0.5, fy negated, z = -1, directions normalised -- data.py:52-70."""
    dev = cameras.device
    center = torch.tensor([K.cx, K.cy], dtype=torch.float, device=dev)
    focal = torch.tensor([K.fx, -K.fy], dtype=torch.float, device=dev)
    xs, ys = torch.meshgrid(torch.arange(K.w, dtype=torch.float, device=dev),
                            torch.arange(K.h, dtype=torch.float, device=dev), indexing="xy")
    grid = (torch.stack([xs, ys], -1) - center + 0.5) / focal
    grid = torch.nn.functional.pad(grid, (0, 1), "constant", -1.)
    R, t = cameras[:, :3, :3], cameras[:, :3, 3]
    d = torch.einsum("hwc,nkc->nhwk", grid, R)
    d = d / torch.norm(d, dim=-1, keepdim=True)
    o = t[:, None, None, :].expand_as(d)
    return o.contiguous(), d.contiguous()


def look_at_origin_poses(n_views: int, radius: float = 4.0311, seed: int = 0, device: str = "cpu") -> torch.Tensor:
    """Blender-synthetic style cameras: on a sphere of `radius`, upper hemisphere, looking at the origin
    (camera looks down its -z axis, +y up) -- the pose distribution of SURVEY 8(d)."""
    g = torch.Generator().manual_seed(seed)
    u = torch.rand(n_views, generator=g)
    v = torch.rand(n_views, generator=g)
    theta = 2 * math.pi * u
    phi = torch.acos(v * 0.95)                        # elevation: keep off the exact pole
    pos = radius * torch.stack([torch.sin(phi) * torch.cos(theta), torch.sin(phi) * torch.sin(theta), torch.cos(phi)], -1)
    fwd = -pos / pos.norm(dim=-1, keepdim=True)       # viewing direction
    up = torch.tensor([0., 0., 1.]).expand_as(fwd)
    right = torch.cross(fwd, up, dim=-1)
    right = right / right.norm(dim=-1, keepdim=True)
    cam_up = torch.cross(right, fwd, dim=-1)
    c2w = torch.eye(4).repeat(n_views, 1, 1)
    c2w[:, :3, 0] = right
    c2w[:, :3, 1] = cam_up
    c2w[:, :3, 2] = -fwd
    c2w[:, :3, 3] = pos
    return c2w.to(device)


def blender_intrinsics(res: int = 800, camera_angle_x: float = 0.6911112070083618) -> Intrinsics:
    """data.py:140-142: focal = w / (2 tan(angle/2)); 800x800 -> 1111.111."""
    focal = res / (2.0 * math.tan(0.5 * camera_angle_x))
    return Intrinsics(focal, focal, res / 2.0, res / 2.0, res, res)


def synthetic_scene(n_views: int = 100, res: int = 800, seed: int = 0, device: str = "cuda",
                    with_colors: bool = True) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor], Intrinsics, torch.Tensor]:
    """Flat ray tables [M,3] for `n_views` synthetic 800x800 cameras plus smooth pseudo-colours.

    Colours are an analytic function of the ray (a shaded ball in front of the white background) so that
    training has a signal without any dataset on disk."""
    K = blender_intrinsics(res)
    cams = look_at_origin_poses(n_views, seed=seed, device=device)
    o, d = generate_rays(cams, K)
    o, d = o.reshape(-1, 3), d.reshape(-1, 3)
    rgbs = None
    if with_colors:
        # ray / sphere(radius .75) intersection -> lambert-ish colour by hit position, else white
        b = (o * d).sum(-1)
        c = (o * o).sum(-1) - 0.75 ** 2
        disc = b * b - c
        hit = disc > 0
        t = -b - torch.sqrt(disc.clamp_min(0))
        p = o + d * t[:, None]
        col = 0.5 + 0.5 * torch.sin(4.0 * p)
        rgbs = torch.where(hit[:, None], col, torch.ones_like(col)).contiguous()
    return o, d, rgbs, K, cams
