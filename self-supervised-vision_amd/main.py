"""Command line of the accelerated path.

The flag surface is the reference's (main.py:11-12,38-43), verbatim: -c/--config, -m/--arch, -a/--algo, -t/--task,
-o/--output, -l/--load with the same choices, so scripts written for the reference keep working.  Algorithms outside the
accelerated two-view path (pirl, deep_cluster, swav, sela) stay on the
surface and raise NotImplementedError.  Multi-GPU: `SSV_GPUS=N python main.py ...` (the repo-root script starts its N ranks itself,
ssv_amd/launch.py) or `python -m torch.distributed.run --nproc-per-node N main.py ...`.
"""
import argparse
import importlib
import os
import time

import numpy as np

TASKS = ("train", "linear_eval", "get_features")
NETWORKS = ("resnet18", "resnet50", "resnext50", "resnext101", "wide_resnet50", "wide_resnet101", "vit")
# algo -> (module, class) for what is built; None marks flag values that exist but are not accelerated
ALGORITHMS = {"simclr": ("simclr", "SimCLR"), "moco": ("moco", "MoCo"), "byol": ("byol", "BYOL"), "dino": ("dino", "DINO"), "pirl": None,
              "barlow": ("barlow", "BarlowTwins"), "simsiam": ("simsiam", "SimSiam"), "relic": ("relic", "ReLIC"), "deep_cluster": None, "swav": None, "sela": None}

_FLAGS = (
    ("-c", "--config", dict(required=True, help="YAML configuration file")),
    ("-m", "--arch", dict(required=True, choices=NETWORKS, help="encoder architecture")),
    ("-a", "--algo", dict(required=True, choices=tuple(ALGORITHMS), help="self-supervised algorithm")),
    ("-t", "--task", dict(required=True, choices=TASKS, help="what to run")),
    ("-o", "--output", dict(default=None, help="name of the output directory (default: a timestamp)")),
    ("-l", "--load", dict(default=None, help="directory holding a best_model.pt to load")),
)


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for short, long_, kw in _FLAGS:
        ap.add_argument(short, long_, type=str, **kw)
    args = vars(ap.parse_args(argv))
    if args["output"] is None:
        args["output"] = time.strftime("%d-%m-%Y_%H-%M")
    return args


def trainer_class(algo):
    entry = ALGORITHMS[algo]
    if entry is None:
        built = ", ".join(k for k, v in ALGORITHMS.items() if v)
        raise NotImplementedError(f"--algo {algo} is not on the accelerated path (built: {built})")
    module, name = entry
    return getattr(importlib.import_module(f"{__package__}.models.{module}"), name)


def main(argv=None):
    args = parse(argv)
    if (args["arch"] == "vit") != (args["algo"] == "dino"):
        raise NotImplementedError("--arch vit and --algo dino go together: DINO is built on the ViT encoder, the two-view algorithms on ResNets")
    if args["task"] != "train" and args["load"] is None:
        raise NotImplementedError("For inference tasks, model checkpoint must be specified using --load")
    model = trainer_class(args["algo"])(args=args)
    if args["task"] == "train":
        model.train()
    elif args["task"] == "linear_eval":
        model.perform_linear_eval()
    else:   # get_features: <split>_fvecs.npy / <split>_gt.npy in the run directory (binary files; the reference opens them in text mode)
        for split in ("train", "test"):
            fvecs, gt = model.build_features(split=split)
            np.save(os.path.join(model.output_dir, f"{split}_fvecs.npy"), fvecs)
            np.save(os.path.join(model.output_dir, f"{split}_gt.npy"), gt)
    return model


if __name__ == "__main__":
    main()
