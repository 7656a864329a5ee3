"""BYOL on the HIP path - drop-in for the reference trainer (models/byol.py:62-208).

Reference quirks kept on purpose: the target network is an independently initialised
copy (not a clone of the online one), runs in train mode (batch statistics, own running stats),
the EMA walks ``zip(online.parameters(), target.parameters())`` (encoder + projector align, the
predictor is dropped, buffers are untouched) and tau follows the within-epoch step index."""
import math

import torch

from .. import _lib, nn as hnn, ops
from ..utils import losses, train_utils
from .base import NETWORKS, TwoViewTrainer
from .heads import ByolMLP as MLP  # noqa: F401


class OnlineNetwork(hnn.HipModule):
    def __init__(self, encoder, encoder_dim, projection_dim):
        super().__init__()
        self.encoder = encoder
        self.proj_head = MLP(encoder_dim, projection_dim)
        self.pred_head = MLP(projection_dim, projection_dim)

    def _prepare_input(self, x):
        return self.encoder._prepare_input(x)

    def _run(self, tape, x):
        x = self.pred_head._run(tape, self.proj_head._run(tape, self.encoder._run(tape, x)))
        return hnn.l2_normalize(tape, x)


class TargetNetwork(hnn.HipModule):
    def __init__(self, encoder, encoder_dim, projection_dim):
        super().__init__()
        self.encoder = encoder
        self.proj_head = MLP(encoder_dim, projection_dim)

    def _prepare_input(self, x):
        return self.encoder._prepare_input(x)

    def _run(self, tape, x):
        return hnn.l2_normalize(tape, self.proj_head._run(tape, self.encoder._run(tape, x)))


class BYOL(TwoViewTrainer):
    algo = "byol"
    graph_safe = True    # the step holds no per-step host state: inputs, loss, BatchNorm statistics, optimizer state are device memory
    graph_inputs = ("aug_1", "aug_2")

    def _build(self, arch):
        encoder, encoder_dim = NETWORKS[arch].values()
        self.online_network = OnlineNetwork(encoder(**self.config["encoder"]), encoder_dim, self.config["proj_dim"]).to(self.device)
        self.target_network = TargetNetwork(encoder(**self.config["encoder"]), encoder_dim, self.config["proj_dim"]).to(self.device)
        self.max_steps = self.config["epochs"] * len(self.train_loader)
        self.tau = self.config.get("tau", 0.996)
        for p in self.target_network.parameters():
            p.requires_grad = False
        self.optim = train_utils.get_optimizer(self.config["optimizer"], params=self.online_network.parameters())
        # the target's tensors go into an arena with the SAME offsets as the online prefix -> one EMA launch
        self._target_arena = train_utils.ParamArena(list(self.target_network.parameters()), with_grads=False)
        self.loss_fn = losses.byol_pair_loss

    def update_tau(self, step):
        tau_upper, tau_lower = self.config.get("tau_upper", 1.0), self.config.get("tau_lower", 0.996)
        self.tau = tau_upper - (tau_upper - tau_lower) * (math.cos(math.pi * step / self.max_steps) + 1) / 2

    @torch.no_grad()
    def momentum_update(self):
        ops.ema_(self._target_arena.data, self.optim.arena.data, self.tau)

    def _after_step(self, step):
        self.update_tau(step)
        self.momentum_update()

    def _embed(self, img):
        return self.online_network(img)

    def _features(self, img):
        return self.online_network(img)          # already L2-normalised (models/byol.py:47)

    def train_step(self, batch):
        img_1, img_2 = batch["aug_1"].to(self.device), batch["aug_2"].to(self.device)
        with hnn.parallel_views(self.device) as pv:
            with pv.view(0):
                with torch.no_grad():            # target params have requires_grad=False: no graph in the reference either
                    target_1 = self.target_network(img_1)
                online_1 = self.online_network(img_1)
            with pv.view(1):
                with torch.no_grad():
                    target_2 = self.target_network(img_2)
                online_2 = self.online_network(img_2)
        loss = self.loss_fn(online_1, online_2, target_1, target_2)
        loss_now = hnn.early_item(loss)                  # the scalar leaves for the host now; the backward does not wait for it, nor it for the backward
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        return {"loss": loss_now.get()}

    def _checkpoint_state(self):
        return {"encoder": self.online_network.state_dict()}

    def _load_state(self, state):
        self.online_network.load_state_dict(state["encoder"])
