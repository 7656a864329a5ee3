"""Shared trainer skeleton for the two-view algorithms (SimCLR / BYOL / Barlow Twins).

The reference repeats this ~170-line class per algorithm (models/simclr.py:39-167 and siblings);
here it is written once.  Public surface kept: ``Cls(args: dict)``, ``train()``,
``train_step(batch) -> {"loss": float}``, ``knn_validate()``, ``build_features(split)``,
``perform_linear_eval()``, ``save_checkpoint()``, ``load_checkpoint(dir)``, ``adjust_learning_rate(epoch)``
and the attributes ``config, output_dir, logger, device, optim, scheduler, loss_fn, best_metric``.
"""
import os

import numpy as np
import torch

from ..networks import resnet
from ..utils import common, data_utils, eval_utils, train_utils
from .. import distributed as hdist

try:                                    # observability only; absent on air-gapped boxes
    import wandb as _wandb
except Exception:                       # pragma: no cover
    _wandb = None

NETWORKS = {
    "resnet18": {"net": resnet.resnet18, "dim": 512},
    "resnet50": {"net": resnet.resnet50, "dim": 2048},
    "resnext50": {"net": resnet.resnext50_32x4d, "dim": 2048},
    "resnext101": {"net": resnet.resnext101_32x8d, "dim": 2048},
    "wide_resnet50": {"net": resnet.wide_resnet50_2, "dim": 2048},
    "wide_resnet101": {"net": resnet.wide_resnet101_2, "dim": 2048},
}


class TwoViewTrainer:
    algo = None          # "simclr" | "byol" | "barlow": names the outputs/<algo>/ directory

    def __init__(self, args):
        assert args["arch"] in NETWORKS.keys(), f"Expected 'arch' to be one of {list(NETWORKS.keys())}"
        output_root = os.path.join("outputs", self.algo, args["arch"])
        self.config, self.output_dir, self.logger, self.device = common.initialize_experiment(args, output_root)
        self.train_loader, self.test_loader = data_utils.get_double_augment_dataloaders(**self.config["data"], device=self.device)
        self._wandb = None
        if _wandb is not None and self.config.get("wandb") and os.environ.get("WANDB_MODE") != "disabled" and hdist.rank() == 0:
            try:
                run = _wandb.init(**self.config["wandb"])
                self.logger.write("Wandb url: {}".format(run.get_url()), mode="info")
                self._wandb = _wandb
            except Exception as e:      # no network: keep training
                self.logger.write(f"wandb disabled: {e}", mode="info")
        self._build(args["arch"])
        self.scheduler, self.warmup_epochs = train_utils.get_scheduler(
            {**self.config["scheduler"], "epochs": self.config["epochs"]}, optimizer=self.optim)
        if self.warmup_epochs > 0:
            self.warmup_rate = (self.config["optimizer"]["lr"] - 1e-12) / self.warmup_epochs
        hdist.attach_grad_sync(self.optim)
        self.best_metric = 0
        if args["load"] is not None:
            self.load_checkpoint(args["load"])

    # -- to be provided by the algorithm -------------------------------------------------------
    def _build(self, arch):
        raise NotImplementedError

    def _embed(self, img):
        raise NotImplementedError

    def _checkpoint_state(self):
        raise NotImplementedError

    def _load_state(self, state):
        raise NotImplementedError

    def _after_step(self, step):
        pass

    # -- shared -------------------------------------------------------------------------------------
    def _log(self, payload):
        if self._wandb is not None:
            self._wandb.log(payload)

    def save_checkpoint(self):
        if hdist.rank() == 0:
            torch.save(self._checkpoint_state(), os.path.join(self.output_dir, "best_model.pt"))

    def load_checkpoint(self, ckpt_dir):
        path = os.path.join(ckpt_dir, "best_model.pt")          # the file save_checkpoint writes
        if not os.path.exists(path):
            raise NotImplementedError(f"Could not find saved checkpoint at {ckpt_dir}")
        self._load_state(torch.load(path, map_location=self.device))
        self.logger.print(f"Successfully loaded model from {ckpt_dir}")

    def adjust_learning_rate(self, epoch):
        if epoch <= self.warmup_epochs:
            for group in self.optim.param_groups:
                group["lr"] = 1e-12 + epoch * self.warmup_rate
        elif self.scheduler is not None:
            self.scheduler.step()

    @torch.no_grad()
    def build_features(self, split="train"):
        if split not in ("train", "test"):
            raise ValueError(f"Unrecognized split {split}, expected one of [train, test]")
        loader = self.train_loader if split == "train" else self.test_loader
        fvecs, gt = [], []
        for step, batch in enumerate(loader):
            z = self._features(batch["img"].to(self.device))
            fvecs.append(z.detach().cpu().numpy())
            gt.append(batch["label"].detach().cpu().numpy())
            common.progress_bar(progress=(step + 1) / len(loader), desc=f"Building {split} features")
        print()
        return np.concatenate(fvecs, axis=0), np.concatenate(gt, axis=0)

    def _features(self, img):
        from .. import ops
        z = self._embed(img)
        return ops.l2norm_fwd(z.contiguous(), normalize=True)[0]

    @torch.no_grad()
    def knn_validate(self):
        fvecs, gt = self.build_features(split="test")
        return eval_utils.compute_neighbor_accuracy(fvecs, gt)

    def perform_linear_eval(self):
        train_vecs, train_gt = self.build_features(split="train")
        test_vecs, test_gt = self.build_features(split="test")
        acc = eval_utils.linear_evaluation(
            config=self.config["linear_eval"], train_data={"fvecs": train_vecs, "labels": train_gt},
            test_data={"fvecs": test_vecs, "labels": test_gt}, num_classes=10, device=self.device)
        self.logger.write("Test linear eval accuracy: {:.4f}".format(acc), mode="info")

    def train(self):
        self.logger.print("Beginning training.", mode="info")
        epochs = self.config["epochs"]
        for epoch in range(1, epochs + 1):
            meter = common.AverageMeter()
            desc = "[TRAIN] Epoch {:4d}/{:4d}".format(epoch, epochs)
            for step, batch in enumerate(self.train_loader):
                metrics = self.train_step(batch)
                self._log({"Train loss": metrics["loss"]})
                meter.add(metrics)
                if hdist.rank() == 0:
                    common.progress_bar(progress=(step + 1) / len(self.train_loader), desc=desc, status=meter.return_msg())
                self._after_step(step)
            print()
            self.logger.write("Epoch {:4d}/{:4d} ".format(epoch, epochs) + meter.return_msg(), mode="train")
            self.adjust_learning_rate(epoch)
            if epoch % self.config["eval_every"] == 0:
                knn_acc = self.knn_validate()
                self.logger.record("Epoch {:4d}/{:4d} [accuracy] {:.4f}".format(epoch, epochs, knn_acc), mode="val")
                self._log({"KNN accuracy": knn_acc, "Epoch": epoch})
                if knn_acc > self.best_metric:
                    self.best_metric = knn_acc
                    self.save_checkpoint()
        print()
        self.logger.print("Completed training. Beginning linear evaluation.", mode="info")
        self.perform_linear_eval()
