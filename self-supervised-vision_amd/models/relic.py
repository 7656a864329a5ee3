"""ReLIC on the HIP path - drop-in for the reference trainer (models/relic.py:23-135,186-193).

Five encoder passes per step like the reference: the online network on both views and on the un-augmented image, the target
network (EMA of the online one, tau schedule on the in-epoch step index) on both views; loss = RelicLoss(o1, t2, orig) +
RelicLoss(o2, t1, orig).  The two view passes run on two HIP streams; the third online pass follows on the ambient stream, so
every BatchNorm sees the views in the reference's order (aug_1, aug_2, img)."""
import math

import torch

from .. import _lib, nn as hnn, ops
from ..utils import losses, train_utils
from .base import NETWORKS, TwoViewTrainer
from .byol import OnlineNetwork, TargetNetwork          # same MLP heads (models/relic.py:23-59 == models/byol.py:24-59)


class ReLIC(TwoViewTrainer):
    algo = "relic"
    graph_safe = True    # the step holds no per-step host state (tau / EMA run between steps, eagerly)
    graph_inputs = ("img", "aug_1", "aug_2")

    def _build(self, arch):
        encoder, encoder_dim = NETWORKS[arch].values()
        cfg = self.config
        self.online_network = OnlineNetwork(encoder(**cfg["encoder"]), encoder_dim, cfg["proj_dim"]).to(self.device)
        self.target_network = TargetNetwork(encoder(**cfg["encoder"]), encoder_dim, cfg["proj_dim"]).to(self.device)
        self.max_steps = cfg["epochs"] * len(self.train_loader)
        self.tau = cfg.get("tau", 0.996)
        for p in self.target_network.parameters():
            p.requires_grad = False
        self.optim = train_utils.get_optimizer(cfg["optimizer"], params=self.online_network.parameters())
        self._target_arena = train_utils.ParamArena(list(self.target_network.parameters()), with_grads=False)
        self.loss_fn = losses.RelicLoss(**cfg["loss_fn"])

    def update_tau(self, step):
        upper, lower = self.config.get("tau_upper", 1.0), self.config.get("tau_lower", 0.996)
        self.tau = upper - (upper - lower) * (math.cos(math.pi * step / self.max_steps) + 1) / 2

    @torch.no_grad()
    def momentum_update(self):
        ops.ema_(self._target_arena.data, self.optim.arena.data, self.tau)

    def _after_step(self, step):
        self.update_tau(step)
        self.momentum_update()

    def _embed(self, img):
        return self.online_network(img)

    def _features(self, img):
        return self.online_network(img)

    def train_step(self, batch):
        img_orig, img_1, img_2 = (batch[k].to(self.device) for k in ("img", "aug_1", "aug_2"))
        with hnn.parallel_views(self.device) as pv:
            with pv.view(0):
                online_1 = self.online_network(img_1)
                with torch.no_grad():
                    target_1 = self.target_network(img_1)
            with pv.view(1):
                online_2 = self.online_network(img_2)
                with torch.no_grad():
                    target_2 = self.target_network(img_2)
        # the third pass (the un-augmented image) accumulates its parameter gradients into view 0's slab: it runs on view 0's STREAM, behind both views (the
        # reference's order of the BatchNorm running-statistics updates), so that its backward is serialised with view 0's on that stream - on the ambient stream
        # the two backward passes walked the same layers at the same time and both read-modified-wrote the same gradient slab (found by the step-graph test:
        # replays summed in another order than the eager run)
        with hnn.parallel_views(self.device) as pv3:
            with pv3.view(0):
                orig_features = self.online_network(img_orig)
        loss = self.loss_fn(online_1, target_2, orig_features) + self.loss_fn(online_2, target_1, orig_features)
        loss_now = hnn.early_item(loss)                  # the scalar leaves for the host now; the backward does not wait for it, nor it for the backward
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        return {"loss": loss_now.get()}

    def _checkpoint_state(self):
        return {"encoder": self.online_network.state_dict()}

    def _load_state(self, state):
        self.online_network.load_state_dict(state["encoder"])
