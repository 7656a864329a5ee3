"""Trainer skeleton shared by the two-view algorithms (SimCLR / BYOL / Barlow Twins).

The reference carries one ~170-line trainer per algorithm (models/simclr.py:39-167 and its siblings); on this path the
common part exists once and an algorithm supplies four hooks (`_build`, `_embed`, `_checkpoint_state`, `_load_state`)
plus `train_step`.  What callers of the reference rely on is kept: ``Cls(args: dict)``; ``train()``,
``train_step(batch) -> {"loss": float}``, ``knn_validate()``, ``build_features(split)``, ``perform_linear_eval()``,
``save_checkpoint()``, ``load_checkpoint(dir)``, ``adjust_learning_rate(epoch)``; attributes ``config, output_dir, logger,
device, optim, scheduler, loss_fn, best_metric``; the log-line formats; ``best_model.pt`` in the run directory.
"""
import os

import numpy as np
import torch

from .. import distributed as hdist
from ..networks import resnet
from ..utils import common, data_utils, eval_utils, train_utils

_WARMUP_FLOOR = 1e-12       # learning rate of "epoch 0" in the linear warm-up


def _encoder_row(factory, dim):
    return {"net": factory, "dim": dim}


# --arch value -> constructor and feature width (same keys and row shape as the reference tables)
NETWORKS = {name: _encoder_row(getattr(resnet, factory), dim) for name, factory, dim in (
    ("resnet18", "resnet18", 512), ("resnet50", "resnet50", 2048),
    ("resnext50", "resnext50_32x4d", 2048), ("resnext101", "resnext101_32x8d", 2048),
    ("wide_resnet50", "wide_resnet50_2", 2048), ("wide_resnet101", "wide_resnet101_2", 2048))}


class _Tracker:
    """wandb when it is importable, reachable and this is rank 0; otherwise a no-op (GPU boxes here have no network)."""

    def __init__(self, settings, logger):
        self._run = None
        if not settings or os.environ.get("WANDB_MODE") == "disabled" or hdist.rank() != 0:
            return
        try:
            import wandb
            run = wandb.init(**settings)
            logger.write("Wandb url: {}".format(run.get_url()), mode="info")
            self._run = wandb
        except Exception as exc:
            logger.write(f"wandb disabled: {exc}", mode="info")

    def log(self, payload):
        if self._run is not None:
            self._run.log(payload)


class TwoViewTrainer:
    algo = None          # "simclr" | "byol" | "barlow" | "dino": names the outputs/<algo>/ directory
    graph_safe = False   # may train_step be replayed as a HIP graph (graph.StepGraph)?  Only steps whose every per-step quantity lives in device memory
    archs = tuple(NETWORKS)

    def __init__(self, args):
        arch = args["arch"]
        if arch not in self.archs:
            raise AssertionError(f"Expected 'arch' to be one of {list(self.archs)}")
        run_root = os.path.join("outputs", self.algo, arch)
        self.config, self.output_dir, self.logger, self.device = common.initialize_experiment(args, run_root)
        self.train_loader, self.test_loader = self._make_loaders()
        self._tracker = _Tracker(self.config.get("wandb"), self.logger)
        self._build(arch)                                               # encoder, heads, self.optim, self.loss_fn
        sched_cfg = dict(self.config["scheduler"], epochs=self.config["epochs"])
        self.scheduler, self.warmup_epochs = train_utils.get_scheduler(sched_cfg, optimizer=self.optim)
        if self.warmup_epochs > 0:
            self.warmup_rate = (self.config["optimizer"]["lr"] - _WARMUP_FLOOR) / self.warmup_epochs
        hdist.attach_grad_sync(self.optim, self._sync_modules())
        self.best_metric = 0
        if args["load"] is not None:
            self.load_checkpoint(args["load"])

    def _sync_modules(self):
        """The bridged modules whose parameters the optimizer owns: their stages are the buckets of the data-parallel gradient exchange."""
        from .. import nn as hnn
        owned = {id(p) for p in self.optim.arena.params}
        return [m for m in vars(self).values() if isinstance(m, hnn.HipModule) and any(id(p) in owned for p in m.parameters())]

    # ---- hooks an algorithm fills in ------------------------------------------------------------------------------
    def _build(self, arch):
        raise NotImplementedError

    def _embed(self, img):
        raise NotImplementedError

    def _checkpoint_state(self):
        raise NotImplementedError

    def _load_state(self, state):
        raise NotImplementedError

    def _make_loaders(self):
        return data_utils.get_double_augment_dataloaders(**self.config["data"], device=self.device)

    def _after_epoch(self, epoch):
        """Per-epoch schedules other than the learning rate (DINO: teacher EMA, weight decay, teacher temperature)."""

    def _after_step(self, step):
        """Called after every optimiser step with the within-epoch step index (BYOL: tau schedule + EMA)."""

    def _features(self, img):
        """Evaluation embedding: projector output on the unit sphere."""
        from .. import ops
        return ops.l2norm_fwd(self._embed(img).contiguous(), normalize=True)[0]

    # ---- checkpoints ----------------------------------------------------------------------------------------------
    def _checkpoint_path(self, directory):
        return os.path.join(directory, "best_model.pt")

    def save_checkpoint(self):
        if hdist.rank() == 0:
            torch.save(self._checkpoint_state(), self._checkpoint_path(self.output_dir))

    def load_checkpoint(self, ckpt_dir):
        path = self._checkpoint_path(ckpt_dir)
        if not os.path.exists(path):
            raise NotImplementedError(f"Could not find saved checkpoint at {ckpt_dir}")
        self._load_state(torch.load(path, map_location=self.device))
        from .. import ops
        ops.invalidate_weight_caches()                                  # parameters were overwritten in place
        self.logger.print(f"Successfully loaded model from {ckpt_dir}")

    # ---- schedule -------------------------------------------------------------------------------------------------
    def adjust_learning_rate(self, epoch):
        """Linear warm-up by epoch, then one scheduler step per epoch (reference models/simclr.py:96-101)."""
        if epoch > self.warmup_epochs:
            if self.scheduler is not None:
                self.scheduler.step()
            return
        lr = _WARMUP_FLOOR + epoch * self.warmup_rate
        for group in self.optim.param_groups:
            group["lr"] = lr

    # ---- evaluation -----------------------------------------------------------------------------------------------
    @torch.no_grad()
    def build_features(self, split="train"):
        loaders = {"train": self.train_loader, "test": self.test_loader}
        if split not in loaders:
            raise ValueError(f"Unrecognized split {split}, expected one of [train, test]")
        loader, total = loaders[split], loaders[split].num_eval_batches()
        vecs, labels = [], []
        for done, batch in enumerate(loader.eval_batches(), start=1):      # never sharded: every rank extracts the same features
            vecs.append(self._features(batch["img"].to(self.device)).cpu().numpy())
            labels.append(batch["label"].cpu().numpy())
            if hdist.rank() == 0:
                common.progress_bar(progress=done / total, desc=f"Building {split} features")
        if hdist.rank() == 0:
            print()
        return np.concatenate(vecs), np.concatenate(labels)

    @torch.no_grad()
    def knn_validate(self):
        return eval_utils.compute_neighbor_accuracy(*self.build_features(split="test"))

    def perform_linear_eval(self):
        sets = {}
        for split in ("train", "test"):
            fvecs, labels = self.build_features(split=split)
            sets[split] = {"fvecs": fvecs, "labels": labels}
        acc = eval_utils.linear_evaluation(config=self.config["linear_eval"], train_data=sets["train"], test_data=sets["test"],
                                           num_classes=self.train_loader.num_classes, device=self.device)
        self.logger.write("Test linear eval accuracy: {:.4f}".format(acc), mode="info")

    # ---- training loop --------------------------------------------------------------------------------------------
    def step(self, batch):
        """``train_step(batch)``, replayed as one HIP graph where that pays (graph.StepGraph: small images - the launch-bound regime of the reference's own
        CIFAR configurations - fused SGD, single process; SSV_STEP_GRAPH=0|1|auto).  Same result, same side effects."""
        sg = self.__dict__.get("_step_graph")
        if sg is None:
            from ..graph import StepGraph
            sg = self._step_graph = StepGraph(self, weak=True)
        return sg(batch)

    def _run_epoch(self, tag):
        meter, total = common.AverageMeter(), len(self.train_loader)
        for step, batch in enumerate(self.train_loader):
            metrics = self.step(batch)
            self._tracker.log({"Train loss": metrics["loss"]})
            meter.add(metrics)
            if hdist.rank() == 0:
                common.progress_bar(progress=(step + 1) / total, desc=f"[TRAIN] {tag}", status=meter.return_msg())
            self._after_step(step)
        if hdist.rank() == 0:
            print()
        return meter

    def _validate(self, epoch, tag):
        acc = self.knn_validate()
        self.logger.record("{} [accuracy] {:.4f}".format(tag, acc), mode="val")
        self._tracker.log({"KNN accuracy": acc, "Epoch": epoch})
        if acc > self.best_metric:
            self.best_metric = acc
            self.save_checkpoint()

    def train(self):
        self.logger.print("Beginning training.", mode="info")
        last = self.config["epochs"]
        for epoch in range(1, last + 1):
            tag = "Epoch {:4d}/{:4d}".format(epoch, last)
            meter = self._run_epoch(tag)
            self.logger.write(f"{tag} " + meter.return_msg(), mode="train")
            self._after_epoch(epoch)
            self.adjust_learning_rate(epoch)
            if epoch % self.config["eval_every"] == 0:
                self._validate(epoch, tag)
        if hdist.rank() == 0:
            print()
        self.logger.print("Completed training. Beginning linear evaluation.", mode="info")
        self.perform_linear_eval()
