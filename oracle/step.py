"""Oracle (test infrastructure): whole train_step restatements (torch autograd on CPU).

  * SimCLROracle.train_step - SimCLR.train_step, models/simclr.py:86-95
  * BYOLOracle.train_step   - BYOL.train_step + update_tau + momentum_update, models/byol.py:116-135,192-193
  * BarlowOracle.train_step - BarlowTwins.train_step, models/barlow.py:86-95
Construction order (RNG consumption) follows the reference ctors: encoder first, then head(s)
(models/simclr.py:50-52, models/byol.py:75-76, models/barlow.py:50-52).
"""
import torch

from . import nets, losses, optim


def _is_param(key):
    return key.endswith(".weight") or key.endswith(".bias")


def _param_list(d):
    return [v for k, v in d.items() if _is_param(k)]


def _prefixed(prefix, d):
    return {f"{prefix}{k}": v for k, v in d.items()}


class _Base:
    def _setup_optim(self, lr, weight_decay):
        self.lr, self.weight_decay = lr, weight_decay
        self.bufs = [None] * len(self.params)

    def _apply_sgd(self):
        grads = [p.grad for p in self.params]
        optim.sgd_nesterov_step(self.params, grads, self.bufs, self.lr, self.weight_decay)
        for p in self.params:
            p.grad = None


class SimCLROracle(_Base):
    def __init__(self, arch="resnet18", reduce_bottom_conv=True, proj_dim=128, lr=0.2, weight_decay=1e-4,
                 normalize=True, temperature=0.5, seed=420):
        if seed is not None:
            torch.manual_seed(seed)
        self.arch, self.rbc = arch, reduce_bottom_conv
        self.encoder = nets.init_resnet(arch, reduce_bottom_conv)
        self.proj_head = nets.init_simclr_head(nets.ENCODER_DIM[arch], proj_dim)
        self.normalize, self.temperature = normalize, temperature
        self.params = _param_list(self.encoder) + _param_list(self.proj_head)
        for p in self.params:
            p.requires_grad_(True)
        self._setup_optim(lr, weight_decay)

    def embed(self, img):
        return nets.simclr_head_forward(self.proj_head, nets.resnet_forward(self.encoder, img, self.arch, self.rbc))

    def train_step(self, aug_1, aug_2, return_z=False):
        z_1 = self.embed(aug_1)
        z_2 = self.embed(aug_2)
        loss = losses.ntxent_loss(z_1, z_2, self.normalize, self.temperature)
        loss.backward()
        self.last_grads = [p.grad.detach().clone() for p in self.params]
        self._apply_sgd()
        out = {"loss": loss.item()}
        if return_z:
            out["z_1"], out["z_2"] = z_1.detach(), z_2.detach()
        return out

    @torch.no_grad()
    def features(self, img):
        """build_features body, models/simclr.py:109-111 (BN stays in train mode)."""
        return losses.l2_normalize(self.embed(img))

    def state(self):
        return {**_prefixed("encoder.", self.encoder), **_prefixed("proj_head.", self.proj_head)}


class BarlowOracle(_Base):
    def __init__(self, arch="resnet18", reduce_bottom_conv=True, proj_dim=4096, lr=0.02, weight_decay=1.5e-6,
                 normalize=False, off_diagonal_weight=0.005, seed=420):
        if seed is not None:
            torch.manual_seed(seed)
        self.arch, self.rbc = arch, reduce_bottom_conv
        self.encoder = nets.init_resnet(arch, reduce_bottom_conv)
        self.proj_head = nets.init_barlow_head(nets.ENCODER_DIM[arch], proj_dim)
        self.normalize, self.lmbda = normalize, off_diagonal_weight
        self.params = _param_list(self.encoder) + _param_list(self.proj_head)
        for p in self.params:
            p.requires_grad_(True)
        self._setup_optim(lr, weight_decay)

    def embed(self, img):
        return nets.barlow_head_forward(self.proj_head, nets.resnet_forward(self.encoder, img, self.arch, self.rbc))

    def train_step(self, aug_1, aug_2, return_z=False):
        z_1, z_2 = self.embed(aug_1), self.embed(aug_2)
        loss = losses.barlow_loss(z_1, z_2, self.normalize, self.lmbda)
        loss.backward()
        self.last_grads = [p.grad.detach().clone() for p in self.params]
        self._apply_sgd()
        out = {"loss": loss.item()}
        if return_z:
            out["z_1"], out["z_2"] = z_1.detach(), z_2.detach()
        return out

    def state(self):
        return {**_prefixed("encoder.", self.encoder), **_prefixed("proj_head.", self.proj_head)}


class BYOLOracle(_Base):
    def __init__(self, arch="resnet18", reduce_bottom_conv=True, proj_dim=128, lr=0.02, weight_decay=1e-4,
                 tau=0.996, max_steps=1000, seed=420):
        if seed is not None:
            torch.manual_seed(seed)
        self.arch, self.rbc = arch, reduce_bottom_conv
        d = nets.ENCODER_DIM[arch]
        # OnlineNetwork(encoder, proj_head, pred_head) then TargetNetwork(encoder, proj_head): byol.py:75-76
        self.online = {"encoder": nets.init_resnet(arch, reduce_bottom_conv),
                       "proj_head": nets.init_byol_mlp(d, proj_dim), "pred_head": nets.init_byol_mlp(proj_dim, proj_dim)}
        self.target = {"encoder": nets.init_resnet(arch, reduce_bottom_conv), "proj_head": nets.init_byol_mlp(d, proj_dim)}
        self.tau, self.max_steps = tau, max_steps
        self.params = _param_list(self.online["encoder"]) + _param_list(self.online["proj_head"]) + _param_list(self.online["pred_head"])
        self.target_params = _param_list(self.target["encoder"]) + _param_list(self.target["proj_head"])
        for p in self.params:
            p.requires_grad_(True)
        self._setup_optim(lr, weight_decay)

    def online_forward(self, img):
        x = nets.resnet_forward(self.online["encoder"], img, self.arch, self.rbc)
        x = nets.byol_mlp_forward(self.online["pred_head"], nets.byol_mlp_forward(self.online["proj_head"], x))
        return losses.l2_normalize(x)

    @torch.no_grad()
    def target_forward(self, img):
        # target is in train mode too: batch statistics, its own running stats
        x = nets.resnet_forward(self.target["encoder"], img, self.arch, self.rbc)
        return losses.l2_normalize(nets.byol_mlp_forward(self.target["proj_head"], x))

    def train_step(self, aug_1, aug_2, step=None):
        o1, t1 = self.online_forward(aug_1), self.target_forward(aug_1)
        o2, t2 = self.online_forward(aug_2), self.target_forward(aug_2)
        loss = losses.byol_mse_loss(o1, o2, t1, t2)
        loss.backward()
        self.last_grads = [p.grad.detach().clone() for p in self.params]
        self._apply_sgd()
        if step is not None:  # the caller's post-step hooks, models/byol.py:192-193
            self.tau = optim.byol_tau(step, self.max_steps)
            with torch.no_grad():
                optim.ema_update(self.params, self.target_params, self.tau)
        return {"loss": loss.item()}

    def state(self):
        out = {}
        for net, name in ((self.online, "online_network"), (self.target, "target_network")):
            for part, d in net.items():
                out.update(_prefixed(f"{name}.{part}.", d))
        return out


def twin64(make):
    """An fp64 copy of an oracle trainer with the SAME (fp32-drawn) initial weights: the centre both fp32 evaluations - this CPU oracle and
    the HIP path - are measured against where two fp32 runs of a chaotic trajectory cannot be compared with each other."""
    torch.set_default_dtype(torch.float64)
    try:
        m64 = make()
    finally:
        torch.set_default_dtype(torch.float32)
    src = make()

    def copy(dst, s_):
        for k in dst:
            if isinstance(dst[k], dict):
                copy(dst[k], s_[k])
            elif dst[k].dtype.is_floating_point:
                dst[k].data = s_[k].detach().double()
    for attr in ("encoder", "proj_head", "online", "target"):
        if hasattr(m64, attr):
            copy(getattr(m64, attr), getattr(src, attr))
    return m64


def snapshot(m):
    """Weights, momentum buffers (and BYOL's target weights) of an oracle trainer: what a step starts from."""
    snap = {"params": [p.detach().clone() for p in m.params], "bufs": [None if b is None else b.detach().clone() for b in m.bufs]}
    if hasattr(m, "target_params"):
        snap["target"] = [p.detach().clone() for p in m.target_params]
    return snap
