#!/usr/bin/env python3
"""Command line of the reference (train.py:8-46) on the MI355X hot path.

    python train.py --data data/lego --datatype synthetic --output out --method kplanes \\
                    --batch_size 1024 --n_samples 1024 --scene_type aabb
"""
import argparse
import os
import random
import uuid
from pathlib import Path

# (flag, keyword arguments) -- same names, defaults and choices as the reference's parser, plus --max_steps
FLAGS = (
    ("--data", dict(type=str, required=True, help="path to the data folder")),
    ("--datatype", dict(type=str, required=True, choices=["synthetic", "nerfstudio"])),
    ("--output", dict(type=str, required=True, help="path to the output folder")),
    ("--method", dict(type=str, required=True, choices=["vanilla", "kplanes", "cobafa"])),
    ("--scene_type", dict(type=str, default="aabb", choices=["aabb", "unbounded"])),
    ("--batch_size", dict(type=int, default=2048)),
    ("--n_samples", dict(type=int, default=400, help="number of samples per ray")),
    ("--eval", dict(action="store_true")),
    ("--eval_every", dict(type=int, default=None, help="number of train steps between evaluations")),
    ("--eval_n", dict(type=int, default=1, help="number of images to evaluate on")),
    ("--max_steps", dict(type=int, default=None, help="stop early (the recipe's step count is 2048*4096/batch_size)")),
)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(prog="tinynerf", description="Train nerf (MI355X HIP path)")
    for flag, kw in FLAGS:
        parser.add_argument(flag, **kw)
    return parser.parse_args(argv)


def fresh_run_dir(root: Path, args) -> Path:
    """<output>/<8 hex>_<method>_<scene_type>_<n_samples>, never an existing directory (train.py:36-41)."""
    for _ in range(1000):
        candidate = root / f"{uuid.uuid4().hex[:8]}_{args.method}_{args.scene_type}_{args.n_samples}"
        if not candidate.is_dir():
            candidate.mkdir(parents=True)
            return candidate
    raise RuntimeError("could not find a free experiment directory")


def load_split(data, root: Path, split: str, device, rays: bool):
    if not (root / f"transforms_{split}.json").exists():
        return None
    scene = data.parse_nerf_synthetic(root, split)
    return data.RaysDataset(scene, device) if rays else data.PoseDataset(scene, device)


def main(argv=None):
    args = parse_args(argv)
    if args.datatype != "synthetic":
        raise NotImplementedError()                     # as in the reference (train.py:30-31)

    import numpy as np
    import torch
    from tinynerf_amd import data
    from tinynerf_amd.run import TrainConfig, train

    seed = int(os.environ.get("SEED", 0))
    if seed:
        for seeder in (torch.manual_seed, np.random.seed, random.seed):
            seeder(seed)
    device = torch.device("cuda")
    root = Path(args.data)
    train_rays = load_split(data, root, "train", device, rays=True)
    if train_rays is None:
        raise FileNotFoundError(root / "transforms_train.json")
    run_dir = fresh_run_dir(Path(args.output), args)
    print(f"Experiment saved to {run_dir}")
    cfg = TrainConfig(method=args.method, scene_type=args.scene_type, batch_size=args.batch_size, n_samples=args.n_samples, seed=seed)
    train(cfg, train_rays, load_split(data, root, "val", device, rays=False), load_split(data, root, "test", device, rays=False),
          run_dir, args.eval_every, args.eval_n, args.max_steps)


if __name__ == "__main__":
    main()
