"""SimSiam on the HIP path - drop-in for the reference trainer (models/simsiam.py:23-132).

Kept from the reference: the "target" network is an independently initialised network that is NEVER updated (there is no
EMA and no stop-gradient twin in models/simsiam.py - the target simply has requires_grad=False), it runs in train mode
with its own BatchNorm statistics, and the loss is 0.5 * (mean -o1.t2 + mean -o2.t1) on unit vectors."""
import torch

from .. import nn as hnn
from ..utils import losses, train_utils
from .base import NETWORKS, TwoViewTrainer
from .heads import _fresh_linear


class _Relu(torch.nn.Module):
    """Stateless placeholder: keeps the reference's nn.Sequential indices (Linear 0/3/6, BatchNorm 1/4/7)."""


def _proj_head(din, dproj):
    return torch.nn.Sequential(_fresh_linear(din, dproj), hnn.HipBatchNorm(dproj), _Relu(), _fresh_linear(dproj, dproj), hnn.HipBatchNorm(dproj), _Relu(),
                               _fresh_linear(dproj, dproj), hnn.HipBatchNorm(dproj))


def _run_proj(seq, tape, x):
    x = hnn.batchnorm(tape, seq[0]._run(tape, x), seq[1], relu=True)
    x = hnn.batchnorm(tape, seq[3]._run(tape, x), seq[4], relu=True)
    return hnn.batchnorm(tape, seq[6]._run(tape, x), seq[7])


class OnlineNetwork(hnn.HipModule):
    def __init__(self, encoder, encoder_dim, projection_dim, bottleneck_dim):
        super().__init__()
        self.encoder = encoder
        self.proj_head = _proj_head(encoder_dim, projection_dim)
        self.pred_head = torch.nn.Sequential(_fresh_linear(projection_dim, bottleneck_dim), hnn.HipBatchNorm(bottleneck_dim), _Relu(),
                                             _fresh_linear(bottleneck_dim, projection_dim))

    def _prepare_input(self, x):
        return self.encoder._prepare_input(x)

    def _run(self, tape, x):
        x = _run_proj(self.proj_head, tape, self.encoder._run(tape, x))
        x = self.pred_head[3]._run(tape, hnn.batchnorm(tape, self.pred_head[0]._run(tape, x), self.pred_head[1], relu=True))
        return hnn.l2_normalize(tape, x)


class TargetNetwork(hnn.HipModule):
    def __init__(self, encoder, encoder_dim, projection_dim):
        super().__init__()
        self.encoder = encoder
        self.proj_head = _proj_head(encoder_dim, projection_dim)

    def _prepare_input(self, x):
        return self.encoder._prepare_input(x)

    def _run(self, tape, x):
        return hnn.l2_normalize(tape, _run_proj(self.proj_head, tape, self.encoder._run(tape, x)))


class SimSiam(TwoViewTrainer):
    algo = "simsiam"
    graph_safe = True    # the step holds no per-step host state (tau / EMA run between steps, eagerly)
    graph_inputs = ("aug_1", "aug_2")

    def _build(self, arch):
        encoder, encoder_dim = NETWORKS[arch].values()
        cfg = self.config
        self.online_network = OnlineNetwork(encoder(**cfg["encoder"]), encoder_dim, cfg["proj_dim"], cfg["bottleneck_dim"]).to(self.device)
        self.target_network = TargetNetwork(encoder(**cfg["encoder"]), encoder_dim, cfg["proj_dim"]).to(self.device)
        for p in self.target_network.parameters():
            p.requires_grad = False
        self.optim = train_utils.get_optimizer(cfg["optimizer"], params=self.online_network.parameters())
        self.loss_fn = losses.simsiam_pair_loss

    def _embed(self, img):
        return self.online_network(img)

    def _features(self, img):
        return self.online_network(img)              # already unit vectors (models/simsiam.py:46-47)

    def train_step(self, batch):
        img_1, img_2 = batch["aug_1"].to(self.device), batch["aug_2"].to(self.device)
        with hnn.parallel_views(self.device) as pv:
            with pv.view(0):
                online_1 = self.online_network(img_1)
                with torch.no_grad():
                    target_1 = self.target_network(img_1)
            with pv.view(1):
                online_2 = self.online_network(img_2)
                with torch.no_grad():
                    target_2 = self.target_network(img_2)
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
