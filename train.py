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


def main():
    ap = argparse.ArgumentParser(prog="tinynerf", description="Train nerf (MI355X HIP path)")
    ap.add_argument("--data", type=str, required=True, help="path to the data folder")
    ap.add_argument("--datatype", type=str, required=True, choices=["synthetic", "nerfstudio"])
    ap.add_argument("--output", type=str, required=True, help="path to the output folder")
    ap.add_argument("--scene_type", type=str, default="aabb", choices=["aabb", "unbounded"])
    ap.add_argument("--method", type=str, required=True, choices=["vanilla", "kplanes", "cobafa"])
    ap.add_argument("--batch_size", type=int, default=2048)
    ap.add_argument("--n_samples", type=int, default=400, help="number of samples per ray")
    ap.add_argument("--eval", action="store_true")
    ap.add_argument("--eval_every", type=int, default=None, help="number of train steps between evaluations")
    ap.add_argument("--eval_n", type=int, default=1, help="number of images to evaluate on")
    ap.add_argument("--max_steps", type=int, default=None, help="stop early (the recipe's step count is 2048*4096/batch_size)")
    args = ap.parse_args()

    import numpy as np
    import torch
    from tinynerf_amd import data
    from tinynerf_amd.run import TrainConfig, train

    seed = int(os.environ.get("SEED", 0))
    if seed != 0:
        torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)
    if args.datatype != "synthetic":
        raise NotImplementedError()                     # as in the reference (train.py:30-31)
    dev = torch.device("cuda")
    root = Path(args.data)
    train_rays = data.RaysDataset(data.parse_nerf_synthetic(root, "train"), dev)
    eval_set = data.PoseDataset(data.parse_nerf_synthetic(root, "val"), dev) if (root / "transforms_val.json").exists() else None
    test_set = data.PoseDataset(data.parse_nerf_synthetic(root, "test"), dev) if (root / "transforms_test.json").exists() else None
    out = Path(args.output)
    while True:
        name = f"{str(uuid.uuid4())[:8]}_{args.method}_{args.scene_type}_{args.n_samples}"
        if not (out / name).is_dir():
            break
    out = out / name
    out.mkdir(parents=True)
    print(f"Experiment saved to {out}")
    cfg = TrainConfig(method=args.method, scene_type=args.scene_type, batch_size=args.batch_size, n_samples=args.n_samples, seed=seed)
    train(cfg, train_rays, eval_set, test_set, out, args.eval_every, args.eval_n, args.max_steps)


if __name__ == "__main__":
    main()
