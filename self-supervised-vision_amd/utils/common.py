"""Experiment plumbing around the accelerated path.

Behavioural contract taken from the reference (utils/common.py): seed 420 everywhere, the YAML config is loaded
as nested dicts and echoed to stdout, every run owns outputs/<algo>/<arch>/<name>/ with ``trainlogs.txt`` and
``hyperparameters.txt``; meters report running means as ``[key] value`` pairs; the progress line is redrawn in place.
Written independently: meters keep running sums, the logger owns its file handle, and there is no CPU device - the
path needs the GPU and says so.
"""
import os
import random
import sys
import time

import numpy as np
import torch
import yaml

_ANSI = dict(info="\x1b[33m", val="\x1b[94m", bar="\x1b[32m", off="\033[0m")
_TAGS = dict(info="[INFO] ", train="[TRAIN] ", val="[VALID] ")


class AverageMeter:
    """Running mean per metric name."""

    def __init__(self):
        self.reset()

    def reset(self):
        self._sum, self._n = {}, {}

    def add(self, metrics):
        for name, value in metrics.items():
            self._sum[name] = self._sum.get(name, 0.0) + float(value)
            self._n[name] = self._n.get(name, 0) + 1

    @property
    def metrics(self):
        return {k: self._sum[k] / self._n[k] for k in self._sum}

    def return_dict(self):
        return self.metrics

    def return_msg(self):
        return "".join(f"[{k}] {v:.4f} " for k, v in self.metrics.items())


class Logger:
    """Tagged lines to <output_dir>/trainlogs.txt (write) and / or the terminal (print); record does both.
    ``active=False`` (every data-parallel rank but 0) makes it a no-op: one process owns the run directory and the terminal."""

    def __init__(self, output_dir, active=True):
        self._fh = open(os.path.join(output_dir, "trainlogs.txt"), "a", buffering=1) if active else None

    def print(self, msg, mode=""):
        if self._fh is None:
            return
        tag = _TAGS.get(mode, "")
        colour = _ANSI.get(mode)
        line = f"{tag}{msg}"
        sys.stdout.write((f"{colour}{line}{_ANSI['off']}" if colour else line) + "\n")

    def write(self, msg, mode):
        if self._fh is not None:
            self._fh.write(f"{_TAGS.get(mode, '')}{msg}\n")

    def record(self, msg, mode):
        self.print(msg, mode)
        self.write(msg, mode)


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def progress_bar(progress=0, desc="Progress", status="", barlen=20):
    done = int(round(barlen * progress))
    arrow = "=" * max(done - 1, 0) + ">"
    sys.stdout.write("\r{}: [{}{}{}{}] {:.2f}% {}".format(desc, _ANSI["bar"], arrow, _ANSI["off"], " " * (barlen - done),
                                                       100.0 * progress, status.ljust(30)))
    sys.stdout.flush()


def open_config(file):
    with open(file) as fh:
        return yaml.safe_load(fh)


def seed_everything(seed=420):
    for seeder in (random.seed, np.random.seed, torch.manual_seed):
        seeder(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def initialize_experiment(args, output_root, seed=420):
    """-> (config, output_dir, logger, device); initialises the process group first when launched by torch.distributed.run."""
    if not torch.cuda.is_available():
        raise RuntimeError("self-supervised-vision_amd needs an AMD GPU (MI355X): no HIP device is visible and there is no CPU fallback")
    from .. import distributed as hdist
    hdist.init_from_env()
    seed_everything(seed)
    config = open_config(args["config"])
    # data parallel: rank 0 owns the run directory, the log file and the terminal; its --output name (a per-process timestamp
    # by default) is the one every rank uses, so --load of the run directory finds rank 0's checkpoint
    lead = hdist.rank() == 0
    output_dir = os.path.join(output_root, hdist.broadcast_object(args["output"]))
    dumped = yaml.dump(config)
    if lead:
        os.makedirs(output_dir, exist_ok=True)
        with open(os.path.join(output_dir, "hyperparameters.txt"), "w") as fh:
            fh.write(dumped)
    hdist.barrier()
    logger = Logger(output_dir, active=lead)
    rule = "-" * 40
    logger.print(f"Logging at {output_dir}", mode="info")
    for line in (rule, "{:>20}".format("Configuration"), rule, dumped, rule):
        logger.print(line)
    device = torch.device("cuda", torch.cuda.current_device())
    logger.print(f"Found GPU device: {torch.cuda.get_device_name(device)} ({time.strftime('%Y-%m-%d %H:%M:%S')})", mode="info")
    return config, output_dir, logger, device
