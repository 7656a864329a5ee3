"""SimCLR on the HIP path - drop-in for the reference trainer (models/simclr.py:39-167)."""
from .. import nn as hnn
from ..utils import losses, train_utils
from .base import NETWORKS, TwoViewTrainer
from .heads import SimclrProjectionHead as ProjectionHead  # noqa: F401  (reference name)


class SimCLR(TwoViewTrainer):
    algo = "simclr"
    graph_safe = True    # the step holds no per-step host state: inputs, loss, BatchNorm statistics, optimizer state are device memory
    graph_inputs = ("aug_1", "aug_2")

    def _build(self, arch):
        encoder, encoder_dim = NETWORKS[arch].values()
        self.encoder = encoder(**self.config["encoder"]).to(self.device)               # built first: RNG order
        self.proj_head = ProjectionHead(encoder_dim, self.config["proj_dim"]).to(self.device)
        self.optim = train_utils.get_optimizer(
            self.config["optimizer"], params=list(self.encoder.parameters()) + list(self.proj_head.parameters()))
        self.loss_fn = losses.SimclrLoss(**self.config["loss_fn"])

    def _embed(self, img):
        return self.proj_head(self.encoder(img))

    def train_step(self, batch):
        """Two separate forward passes - BatchNorm statistics are per view, like the reference."""
        img_1, img_2 = batch["aug_1"].to(self.device), batch["aug_2"].to(self.device)
        with hnn.parallel_views(self.device) as pv:      # the two independent passes run on two HIP streams
            with pv.view(0):
                z_1 = self._embed(img_1)
            with pv.view(1):
                z_2 = self._embed(img_2)
        loss = self.loss_fn(z_1, z_2)
        loss_now = hnn.early_item(loss)                  # the scalar leaves for the host now; the backward does not wait for it, nor it for the backward
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        return {"loss": loss_now.get()}

    def _checkpoint_state(self):
        return {"encoder": self.encoder.state_dict(), "proj_head": self.proj_head.state_dict()}

    def _load_state(self, state):
        self.encoder.load_state_dict(state["encoder"])
        self.proj_head.load_state_dict(state["proj_head"])
