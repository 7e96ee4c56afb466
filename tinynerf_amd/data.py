"""Scene loading and ray tables (reference ``src/data.py``).

Same public names as the reference (``Intrinsics``, ``NerfData``, ``PoseDataset``, ``RaysDataset``,
``parse_nerf_synthetic``); the difference is where the rays live: the reference keeps them on the CPU and
feeds them through ``DataLoader(num_workers=8)`` one ray per ``__getitem__`` (run.py:116-122), which cannot
feed >= 1e8 samples/s; here ``generate_rays`` runs on the device and ``RaysDataset`` is three flat HBM tables
that the harness indexes with device-side random indices.
"""
from __future__ import annotations

import json
from dataclasses import dataclass
from pathlib import Path
from typing import List, Optional, Tuple

import numpy as np
import torch

from .rays import Intrinsics, generate_rays as _generate_rays


@dataclass
class NerfData:
    """Images + camera poses (data.py:21-76)."""
    cameras: torch.Tensor                      # [n_images, 4, 4]
    intrinsics: Intrinsics
    imgs: Optional[List[torch.Tensor]] = None  # [n_images][h, w, 3] in [0,1]
    bg_color: Optional[torch.Tensor] = None

    @property
    def n_img(self) -> int:
        return len(self.cameras)

    def generate_rays(self, device: torch.device | str = "cpu") -> Tuple[torch.Tensor, torch.Tensor]:
        """rays_o, rays_d of shape [n_images, h, w, 3] (data.py:48-73), built on `device`."""
        return _generate_rays(self.cameras.to(device), self.intrinsics)

    def scene_scale(self) -> float:
        return torch.max(torch.var(self.cameras[:, :3, 3], 0)).item()      # data.py:75-76


class PoseDataset:
    """Per-image rays for inference (data.py:78-100)."""

    def __init__(self, data: NerfData, device: torch.device | str = "cpu"):
        self.rays_o, self.rays_d = data.generate_rays(device)
        self.rgbs = None if data.imgs is None else [im.to(device) for im in data.imgs]
        self.scene_scale = data.scene_scale()
        self.bg_color = data.bg_color
        self.intrinsics = data.intrinsics

    def img_intrinsics(self, idx: int) -> Intrinsics:
        return self.intrinsics

    def __len__(self) -> int:
        return self.rays_o.size(0)

    def __getitem__(self, idx: int):
        out = {"rays_o": self.rays_o[idx], "rays_d": self.rays_d[idx]}
        if self.rgbs is not None:
            out["rgbs"] = self.rgbs[idx]
        return out


class RaysDataset:
    """All training rays as flat device tables [M,3] (data.py:102-120 without the per-ray __getitem__ path)."""

    def __init__(self, data: NerfData, device: torch.device | str = "cpu"):
        assert data.imgs is not None, "rays datasets requires rgbs"
        o, d = data.generate_rays(device)
        self.rays_o = o.reshape(-1, 3).contiguous()
        self.rays_d = d.reshape(-1, 3).contiguous()
        self.rgbs = torch.cat([im.reshape(-1, 3) for im in data.imgs]).to(device).contiguous()
        self.scene_scale = data.scene_scale()
        self.bg_color = data.bg_color

    def __len__(self) -> int:
        return self.rays_o.size(0)

    def __getitem__(self, idx):
        return {"rays_o": self.rays_o[idx], "rays_d": self.rays_d[idx], "rgbs": self.rgbs[idx]}


def _composite_over(img, bg_color):
    """PIL image -> float32 [h,w,3] in [0,1]; RGBA is composited over `bg_color` first."""
    from PIL import Image
    if img.mode == "RGBA":
        img = Image.alpha_composite(Image.new("RGBA", img.size, bg_color), img).convert("RGB")
    return torch.from_numpy(np.asarray(img, dtype=np.float32) / np.float32(255.))


def parse_nerf_synthetic(scene_path: Path, split: str = "train", bg_color: Tuple[int, int, int] = (255, 255, 255)) -> NerfData:
    """Blender-synthetic scenes (https://www.matthewtancik.com/nerf), data.py:123-158: frames listed in
    ``transforms_<split>.json``, RGBA composited over `bg_color`, one pinhole camera for the whole split with
    focal = w / (2 tan(camera_angle_x / 2)) and the principal point at the image centre."""
    from PIL import Image
    root = Path(scene_path)
    meta = json.loads((root / f"transforms_{split}.json").read_text())
    frames = meta["frames"]
    if not frames:
        raise ValueError(f"{root}: no frames in split {split!r}")
    images = []
    size = None
    for frame in frames:
        with Image.open((root / frame["file_path"]).with_suffix(".png")) as img:
            size = size or img.size
            images.append(_composite_over(img, bg_color))
    width, height = size
    focal = width / (2. * np.tan(0.5 * meta["camera_angle_x"]))
    poses = torch.tensor([frame["transform_matrix"] for frame in frames], dtype=torch.float)
    return NerfData(cameras=poses, intrinsics=Intrinsics(focal, focal, width / 2., height / 2., width, height), imgs=images,
                    bg_color=torch.tensor(bg_color, dtype=torch.float) / 255.)


def parse_nerfstudio(scene_path: Path, split: str = "train", bg_color: Tuple[int, int, int] = (255, 255, 255)) -> NerfData:
    """The reference only stubs this loader (data.py:162-167)."""
    raise NotImplementedError()
