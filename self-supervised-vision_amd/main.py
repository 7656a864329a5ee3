"""Command line of the accelerated path - the reference's flags and choices, verbatim
(main.py:11-12,38-43): -c/--config -m/--arch -a/--algo -t/--task [-o/--output] [-l/--load].

Algorithms outside the accelerated two-view path (moco, dino, pirl, simsiam, relic, deep_cluster,
swav, sela) and the ViT encoder stay on the flag surface but raise NotImplementedError.
Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N main.py ...`.
"""
import argparse
import os
from datetime import datetime as dt

import numpy as np

TASKS = ["train", "linear_eval", "get_features"]
NETWORKS = ["resnet18", "resnet50", "resnext50", "resnext101", "wide_resnet50", "wide_resnet101", "vit"]
ACCELERATED = ("simclr", "byol", "barlow")
ALGO_NAMES = ["simclr", "moco", "byol", "dino", "pirl", "barlow", "simsiam", "relic", "deep_cluster", "swav", "sela"]


def _trainer(algo):
    if algo == "simclr":
        from .models.simclr import SimCLR as cls
    elif algo == "byol":
        from .models.byol import BYOL as cls
    elif algo == "barlow":
        from .models.barlow import BarlowTwins as cls
    else:
        raise NotImplementedError(f"--algo {algo} is not on the accelerated path (built: {', '.join(ACCELERATED)})")
    return cls


def _require_checkpoint(args):
    if args["load"] is None:
        raise NotImplementedError("For inference tasks, model checkpoint must be specified using --load")


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", required=True, type=str, help="Path to configuration file")
    ap.add_argument("-m", "--arch", required=True, type=str, choices=NETWORKS, help="Encoder architecture to use")
    ap.add_argument("-a", "--algo", required=True, type=str, choices=ALGO_NAMES, help="Self-supervised algorithm to work with")
    ap.add_argument("-t", "--task", required=True, type=str, choices=TASKS, help="Task to perform for chosen algorithm")
    ap.add_argument("-o", "--output", default=dt.now().strftime("%d-%m-%Y_%H-%M"), type=str, help="Path to output directory")
    ap.add_argument("-l", "--load", default=None, type=str, help="Path to directory containing trained checkpoints to be loaded")
    args = vars(ap.parse_args(argv))
    if args["arch"] == "vit":
        raise NotImplementedError("--arch vit belongs to DINO, which is not on the accelerated path yet")

    model = _trainer(args["algo"])(args=args)
    task = args["task"]
    if task == "train":
        model.train()
    elif task == "linear_eval":
        _require_checkpoint(args)
        model.perform_linear_eval()
    elif task == "get_features":
        _require_checkpoint(args)
        for split in ("train", "test"):
            fvecs, gt = model.build_features(split=split)
            np.save(os.path.join(model.output_dir, f"{split}_fvecs.npy"), fvecs)      # binary mode (the reference opens "w")
            np.save(os.path.join(model.output_dir, f"{split}_gt.npy"), gt)
    return model


if __name__ == "__main__":
    main()
