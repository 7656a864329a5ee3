"""DINO on the HIP path - drop-in for the reference trainer (models/dino.py:45-241) on its ViT encoder.

Kept from the reference, on purpose (they define the numbers): the teacher is an independently initialised network; the
student's rows are concatenated [global crops ; local crops] and then RE-VIEWED as [batch, views, K] (models/dino.py:151,158 -
this mixes samples, and it is what the loss sees); both teacher global views are scored against every student view of the
other augmented copy (utils/losses.py:80-89); the centre is an EMA of the teacher batch mean updated every step; teacher EMA,
weight decay and teacher temperature move once per EPOCH (:226-229); gradients are clamped element-wise to +-gradient_clip
(:76-79) - here inside the AdamW kernel.
What differs in execution, not in numbers: the two augmented copies' crops of one size go through the network as ONE batch
(the ViT has no batch-coupled layer), so the student runs 2 passes instead of 4 and the teacher 1 instead of 2.
"""
import math

import torch

from .. import _lib, distributed as hdist, nn as hnn, ops
from ..networks import vit
from ..utils import data_utils, train_utils
from .base import TwoViewTrainer
from .heads import _fresh_linear

NETWORKS = {"vit": {"net": vit.TransformerEncoder, "dim": None}}


class _Gelu(torch.nn.Module):
    """Stateless placeholder so proj_head keeps the reference's Sequential indices 0, 2, 4 for its Linear layers."""


class _WeightNormLinear(hnn.HipModule):
    """nn.utils.weight_norm(nn.Linear(din, dout)): parameters registered in the reference order bias, weight_g, weight_v."""

    def __init__(self, din, dout):
        super().__init__()
        lin = _fresh_linear(din, dout)
        v = lin.weight.data
        self.bias = torch.nn.Parameter(lin.bias.data)
        self.weight_g = torch.nn.Parameter(v.norm(dim=1, keepdim=True))
        self.weight_v = torch.nn.Parameter(v)

    def _run(self, tape, x):
        return hnn.weightnorm_linear(tape, x, self.weight_g, self.weight_v, self.bias)


class EncoderModel(hnn.HipModule):
    def __init__(self, encoder, encoder_dim, hidden_dim, projection_dim):
        super().__init__()
        self.encoder = encoder
        self.proj_head = torch.nn.Sequential(_fresh_linear(encoder_dim, hidden_dim), _Gelu(), _fresh_linear(hidden_dim, hidden_dim), _Gelu(),
                                             _fresh_linear(hidden_dim, hidden_dim))
        self.fc_out = _WeightNormLinear(hidden_dim, projection_dim)

    def _prepare_input(self, x):
        return self.encoder._prepare_input(x)

    def _run(self, tape, x):
        x = self.encoder._run(tape, x)
        x = hnn.gelu(tape, self.proj_head[0]._run(tape, x))
        x = hnn.gelu(tape, self.proj_head[2]._run(tape, x))
        x = hnn.l2_normalize(tape, self.proj_head[4]._run(tape, x))
        return self.fc_out._run(tape, x)


class _DinoLossFn(torch.autograd.Function):
    """0.5 * DinoLoss(teacher_1, student_2) + 0.5 * DinoLoss(teacher_2, student_1) and its gradient (two ssv_dino_loss calls).
    sg: student outputs of the global crops, rows [copy 1 ; copy 2]; sl: same for the local crops; tg: teacher outputs."""

    @staticmethod
    def forward(ctx, sg, sl, tg, center, bs, vg, vl, temp_s, temp_t):
        if vg != 2:
            raise NotImplementedError("DinoLoss reads exactly two teacher views per sample (utils/losses.py:83-84): num_global_views must be 2")
        ng, nl, k = bs * vg, bs * vl, sg.shape[1]
        sg, sl, tg = sg.detach(), sl.detach(), tg.detach()
        s1 = torch.cat((sg[:ng], sl[:nl]), 0).view(bs, vg + vl, k)          # the reference's cat-then-view
        s2 = torch.cat((sg[ng:], sl[nl:]), 0).view(bs, vg + vl, k)
        t1, t2 = tg[:ng].view(bs, vg, k), tg[ng:].view(bs, vg, k)
        world = hdist.world_size()
        loss = torch.empty((), dtype=torch.float32, device=sg.device)
        d2 = ops.dino_loss(t1, s2, center, temp_s, temp_t, 0.5 / world, loss, accumulate=False)
        d1 = ops.dino_loss(t2, s1, center, temp_s, temp_t, 0.5 / world, loss, accumulate=True)
        hdist.all_reduce_sum(loss)                                         # global-batch mean on every rank
        d1, d2 = d1.view(-1, k), d2.view(-1, k)
        ctx.saved = (torch.cat((d1[:ng], d2[:ng]), 0), torch.cat((d1[ng:], d2[ng:]), 0))
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dsg, dsl = ctx.saved
        g = dloss.contiguous()
        return ops.scale_(dsg, g), ops.scale_(dsl, g), None, None, None, None, None, None, None


class DINO(TwoViewTrainer):
    algo = "dino"
    graph_safe = True    # every per-step quantity is device memory (AdamW's step count included: ssv_adamw_counted); per-epoch scalars are part of graph_key()
    graph_inputs = ("global_1", "global_2", "local_1", "local_2")
    archs = tuple(NETWORKS)

    def _make_loaders(self):
        return data_utils.get_multicrop_dataloaders(**self.config["data"], device=self.device)

    def _build(self, arch):
        enc_cfg, head_cfg = self.config["encoder"], self.config["proj_head"]
        make = lambda: EncoderModel(NETWORKS[arch]["net"](enc_cfg), enc_cfg["hidden_dim"], head_cfg["hidden_dim"], head_cfg["proj_dim"]).to(self.device)
        self.student_model, self.teacher_model = make(), make()            # student first: RNG order
        self.teacher_center = torch.randn(1, head_cfg["proj_dim"]).to(self.device)
        self.temp_teacher = self.config.get("teacher_temp_lower", 0.04)
        self.temp_student = self.config.get("student_temp", 0.1)
        self.m = self.config.get("center_momentum", 0.9)
        for p in self.teacher_model.parameters():
            p.requires_grad = False
        self.optim = train_utils.get_optimizer(self.config["optimizer"], self.student_model.parameters())
        if self.config.get("gradient_clip", None) is not None:
            if not isinstance(self.optim, train_utils.FusedAdamW):
                raise NotImplementedError("gradient_clip is fused into the AdamW update kernel; optimizer.name must be adamw when it is set")
            self.optim.clip = float(self.config["gradient_clip"])          # the clamp hooks, fused into the update kernel
        self._teacher_arena = train_utils.ParamArena(list(self.teacher_model.parameters()), with_grads=False)
        self.loss_fn = _DinoLossFn.apply

    # ---- per-epoch schedules (models/dino.py:113-134) -------------------------------------------------------------
    def update_temperature(self, epoch):
        lower, upper = self.config.get("teacher_temp_lower", 0.04), self.config.get("teacher_temp_upper", 0.07)
        warm = self.config.get("temp_warmup_epochs", 30)
        self.temp_student = self.config.get("student_temp", 0.1)
        self.temp_teacher = lower + (upper - lower) * (epoch / warm) if epoch <= warm else upper

    def _cosine_ramp(self, epoch, upper, lower):
        return upper - (upper - lower) * (math.cos(math.pi * epoch / self.config["epochs"]) + 1) / 2

    def update_weight_decay(self, epoch):
        wd = self._cosine_ramp(epoch, self.config.get("weight_decay_upper", 0.4), self.config.get("weight_decay_lower", 0.04))
        for group in self.optim.param_groups:
            group["weight_decay"] = wd

    @torch.no_grad()
    def update_teacher_model(self, epoch):
        lbd = self._cosine_ramp(epoch, self.config.get("lambda_upper", 1.0), self.config.get("lambda_lower", 0.996))
        ops.ema_(self._teacher_arena.data, self.optim.arena.data, lbd)

    @torch.no_grad()
    def update_teacher_center(self, teacher_1, teacher_2):
        ops.dino_center_update(self.teacher_center.view(-1), teacher_1, teacher_2, self.m)
        if hdist.is_on():
            hdist.all_reduce_sum(self.teacher_center)
            self.teacher_center.div_(hdist.world_size())

    def _after_epoch(self, epoch):
        self.update_teacher_model(epoch)
        self.update_weight_decay(epoch)
        self.update_temperature(epoch)

    def graph_key(self):
        """Scalars that reach kernels as arguments and move with the epoch schedules: a HIP graph of the step is valid for one value of each (graph.StepGraph)."""
        return (float(self.temp_student), float(self.temp_teacher), float(self.m))

    # ---- the step -------------------------------------------------------------------------------------------------
    def train_step(self, batch):
        g1, g2, l1, l2 = (batch[k].to(self.device) for k in ("global_1", "global_2", "local_1", "local_2"))
        bs, vg = g1.shape[0], g1.shape[1]
        vl = l1.shape[1]
        glob = torch.cat((g1.flatten(0, 1), g2.flatten(0, 1)), 0)          # [2*bs*vg, 3, hg, wg]: both copies' crops in one pass
        loc = torch.cat((l1.flatten(0, 1), l2.flatten(0, 1)), 0)
        with hnn.parallel_views(self.device) as pv:      # independent passes on two HIP streams: MFMA-bound GEMMs of one overlap
            with pv.view(0):                             # the HBM-bound LayerNorm / GELU kernels of the other
                student_g = self.student_model(glob)
            with pv.view(1):
                student_l = self.student_model(loc)
                with torch.no_grad():
                    teacher_g = self.teacher_model(glob)
        loss = self.loss_fn(student_g, student_l, teacher_g, self.teacher_center.view(-1), bs, vg, vl, self.temp_student, self.temp_teacher)
        ng = bs * vg
        self.update_teacher_center(teacher_g[:ng], teacher_g[ng:])
        loss_now = hnn.early_item(loss)                  # the scalar leaves for the host now; the backward does not wait for it, nor it for the backward
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        return {"loss": loss_now.get()}

    def _embed(self, img):
        return self.student_model(img)

    def _features(self, img):
        return self.student_model(img)                                     # build_features uses the raw K-dim output (models/dino.py:181,189)

    def _checkpoint_state(self):
        return {"encoder": self.student_model.state_dict()}

    def _load_state(self, state):
        self.student_model.load_state_dict(state["encoder"])
