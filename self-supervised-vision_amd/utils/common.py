"""Experiment plumbing with the reference's behaviour (utils/common.py): seed 420, YAML config,
output directory with trainlogs.txt + hyperparameters.txt, running-mean meter, progress bar."""
import logging
import os
import random

import numpy as np
import torch
import yaml

_C = {"yellow": "\x1b[33m", "blue": "\x1b[94m", "green": "\x1b[32m", "end": "\033[0m"}


class AverageMeter:
    def __init__(self):
        self.reset()

    def reset(self):
        self.metrics = {}

    def add(self, metrics):
        for key, value in metrics.items():
            self.metrics.setdefault(key, []).append(value)

    def return_dict(self):
        return {key: np.mean(value) for key, value in self.metrics.items()}

    def return_msg(self):
        return "".join("[{}] {:.4f} ".format(k, v) for k, v in self.return_dict().items())


class Logger:
    _PREFIX = {"info": "[INFO] ", "train": "[TRAIN] ", "val": "[VALID] "}

    def __init__(self, output_dir):
        for handler in logging.root.handlers[:]:
            logging.root.removeHandler(handler)
        logging.basicConfig(level=logging.INFO, format="%(message)s",
                            handlers=[logging.FileHandler(os.path.join(output_dir, "trainlogs.txt"))])

    def print(self, msg, mode=""):
        if mode == "info":
            print(f"{_C['yellow']}[INFO] {msg}{_C['end']}")
        elif mode == "train":
            print(f"[TRAIN] {msg}")
        elif mode == "val":
            print(f"{_C['blue']}[VALID] {msg}{_C['end']}")
        else:
            print(f"{msg}")

    def write(self, msg, mode):
        logging.info(self._PREFIX.get(mode, "") + str(msg))

    def record(self, msg, mode):
        self.print(msg, mode)
        self.write(msg, mode)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def progress_bar(progress=0, desc="Progress", status="", barlen=20):
    status = status.ljust(30)
    filled = int(round(barlen * progress))
    bar = _C["green"] + "=" * (filled - 1) + ">" + _C["end"] + " " * (barlen - filled)
    print("\r{}: [{}] {:.2f}% {}".format(desc, bar, progress * 100, status), end="")


def open_config(file):
    with open(file, "r") as f:
        return yaml.safe_load(f)


def seed_everything(seed=420):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def initialize_experiment(args, output_root, seed=420):
    """Returns (config, output_dir, logger, device).  Unlike the reference there is no CPU device:
    the accelerated path needs the GPU and says so."""
    seed_everything(seed)
    config = open_config(args["config"])
    output_dir = os.path.join(output_root, args["output"])
    os.makedirs(output_dir, exist_ok=True)
    logger = Logger(output_dir)
    logger.print("Logging at {}".format(output_dir), mode="info")
    logger.print("-" * 40)
    logger.print("{:>20}".format("Configuration"))
    logger.print("-" * 40)
    logger.print(yaml.dump(config))
    logger.print("-" * 40)
    with open(os.path.join(output_dir, "hyperparameters.txt"), "w") as f:
        f.write(yaml.dump(config))
    if not torch.cuda.is_available():
        raise RuntimeError("self-supervised-vision_amd needs an AMD GPU (MI355X): no HIP device is visible and there is no CPU fallback")
    from .. import distributed as hdist
    hdist.init_from_env()
    device = torch.device("cuda", torch.cuda.current_device())
    logger.print("Found GPU device: {}".format(torch.cuda.get_device_name(device)), mode="info")
    return config, output_dir, logger, device
