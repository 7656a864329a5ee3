"""Optimizer / scheduler factories with the reference's semantics (utils/train_utils.py:6-45),
backed by one fused HIP kernel over a flat parameter arena.

``get_optimizer`` keeps the reference quirks: SGD is always momentum 0.9 + Nesterov (the YAML
``momentum`` / ``nesterov`` keys are ignored) and weight decay hits every tensor.  ``get_scheduler``
keeps the warm-up lr seeding (lr := 1e-12 + lr/warmup_epochs BEFORE the cosine schedule is built).
"""
import torch
import torch.optim.lr_scheduler as lr_scheduler

from .. import _lib, ops

_ALIGN = 64          # floats; every tensor starts on a 256-byte boundary inside the arena


class ParamArena:
    """Packs parameters into ONE contiguous fp32 buffer (and their gradients into another):
    the optimizer update is a single launch, zero_grad a single fill, and the data-parallel
    gradient exchange one contiguous RCCL all-reduce.  ``p.data`` / ``p.grad`` become views that
    keep each parameter's logical shape and memory format (OHWI for conv filters)."""

    def __init__(self, params, with_grads=True):
        self.params = [p for p in params]
        if not self.params:
            raise ValueError("empty parameter list")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise _lib.SsvError("the fused optimizer needs parameters on the GPU (call .to(device) first); no CPU fallback")
        self.offsets, off = [], 0
        for p in self.params:
            if p.dtype != torch.float32 or p.device != dev:
                raise _lib.SsvError("all parameters must be fp32 on one device")
            self.offsets.append(off)
            off += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.numel = off
        self.data = ops.fill_(torch.empty(off, dtype=torch.float32, device=dev), 0.0)
        # two gradient slabs, contiguous: slot 0 = p.grad, slot 1 = p._grad_alt (second view's stream); one fill zeroes both
        self._grads = ops.fill_(torch.empty(2 * off, dtype=torch.float32, device=dev), 0.0) if with_grads else None
        self.grad = self._grads[:off] if with_grads else None
        self.grad_alt = self._grads[off:] if with_grads else None
        for p, o in zip(self.params, self.offsets):
            view = self._view(self.data, p, o)
            view.copy_(p.data)                       # one-time relocation (plumbing)
            p.data = view
            if with_grads:
                p.grad = self._view(self.grad, p, o)
                p._grad_alt = self._view(self.grad_alt, p, o)

    @staticmethod
    def _view(flat, p, off):
        seg = flat[off:off + p.numel()]
        if p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last) and not p.is_contiguous():
            o, i, h, w = p.shape
            return seg.view(o, h, w, i).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def zero_grad(self):
        ops.invalidate_weight_caches()                # weights may have changed since the last backward
        ops.fill_(self._grads, 0.0)


class FusedSGD(torch.optim.Optimizer):
    """optim.SGD(momentum=0.9, nesterov=True, weight_decay=wd) as ONE ssv_sgd_nesterov launch."""

    def __init__(self, params, lr, weight_decay, momentum=0.9):
        params = list(params)
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, momentum=momentum, nesterov=True))
        self.arena = ParamArena(params)
        self.momentum_buffer = ops.fill_(torch.empty_like(self.arena.data), 0.0)
        self._steps = 0
        self.grad_sync = None        # set by the data-parallel wrapper (distributed.attach_grad_sync): its finish() runs before the update
        # (lr, weight decay, momentum, first-step flag) in device memory: what the update of a step CAPTURED as a HIP graph reads (ssv_sgd_nesterov_dev), so that the
        # schedules move four device floats and not kernel arguments - one graph serves the whole run (graph.StepGraph calls push_hyper() before every replay)
        self._hyper_dev = ops.fill_(torch.empty(4, dtype=torch.float32, device=self.arena.data.device), 0.0)
        self._hyper_sent = None

    def hyper(self):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["weight_decay"]), float(g["momentum"]), 0.0 if self._steps > 0 else 1.0)

    def push_hyper(self):
        """Bring the device copy of the hyper-parameters up to date (a blocking 16-byte copy, only when a schedule has moved them)."""
        h = self.hyper()
        if h != self._hyper_sent:
            self._hyper_dev.copy_(torch.tensor(h, dtype=torch.float32))
            self._hyper_sent = h

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        a = self.arena
        from .. import nn as hnn
        hnn.join_view_streams(a.data.device)         # backward kernels of the two view streams must have been ordered before us
        g2 = a.grad_alt
        if self.grad_sync is not None:               # data parallel: the slabs are folded and all-reduced per bucket (distributed.BucketedGradSync)
            self.grad_sync.finish()
            g2 = None
        ops.invalidate_weight_caches()               # the transposed-filter cache describes the weights we are about to change
        if hnn.capturing():                          # recorded into a HIP graph: hyper-parameters from device memory (see __init__)
            _lib.call("ssv_sgd_nesterov_dev", a.numel, _lib.ptr(a.data), _lib.ptr(a.grad), _lib.ptr(g2), _lib.ptr(self.momentum_buffer), _lib.ptr(self._hyper_dev), _lib.stream())
        else:
            _lib.call("ssv_sgd_nesterov", a.numel, _lib.ptr(a.data), _lib.ptr(a.grad), _lib.ptr(g2), _lib.ptr(self.momentum_buffer),
                      float(g["lr"]), float(g["weight_decay"]), float(g["momentum"]), int(self._steps == 0), _lib.stream())
        self._steps += 1


class FusedAdamW(torch.optim.Optimizer):
    """optim.AdamW(lr, weight_decay, eps, betas=(0.9, 0.999)) as ONE ssv_adamw launch over the arena.  ``clip`` > 0 clamps the
    (summed) gradient element-wise first - the reference's DINO trainer does that with tensor hooks (models/dino.py:76-79)."""

    def __init__(self, params, lr, weight_decay, eps=1e-6, betas=(0.9, 0.999), clip=0.0):
        params = list(params)
        super().__init__(params, dict(lr=lr, weight_decay=weight_decay, eps=eps, betas=betas))
        self.arena = ParamArena(params)
        self.exp_avg = ops.fill_(torch.empty_like(self.arena.data), 0.0)
        self.exp_avg_sq = ops.fill_(torch.empty_like(self.arena.data), 0.0)
        self.clip = float(clip)
        self._steps = 0
        # the step count lives in device memory (ssv_adamw_counted): nothing in the update's launch changes from step to step, so the training step can be
        # replayed as a HIP graph (graph.StepGraph); _steps counts the EXECUTED steps on the host (eager steps and replays; StepGraph keeps it right)
        self._step_dev = torch.zeros(1, dtype=torch.int64, device=self.arena.data.device)
        self._bc_dev = ops.fill_(torch.empty(4, dtype=torch.float32, device=self.arena.data.device), 0.0)      # [bc1, 1/sqrt(bc2)] written by the device, [lr, weight decay] by push_hyper()
        self._hyper_sent = None
        self.grad_sync = None

    def hyper(self):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["weight_decay"]))

    def push_hyper(self):
        """Bring the device copy of (lr, weight decay) up to date: what a CAPTURED update reads (ssv_adamw_counted_dev), only when a schedule has moved them."""
        h = self.hyper()
        if h != self._hyper_sent:
            self._bc_dev[2:4].copy_(torch.tensor(h, dtype=torch.float32))
            self._hyper_sent = h

    def zero_grad(self, set_to_none=False):
        self.arena.zero_grad()

    @torch.no_grad()
    def step(self, closure=None):
        g = self.param_groups[0]
        a = self.arena
        from .. import nn as hnn
        hnn.join_view_streams(a.data.device)
        g2 = a.grad_alt
        if self.grad_sync is not None:
            self.grad_sync.finish()
            g2 = None
        self._steps += 1
        ops.invalidate_weight_caches()
        if hnn.capturing():
            _lib.call("ssv_adamw_counted_dev", a.numel, _lib.ptr(a.data), _lib.ptr(a.grad), _lib.ptr(g2), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                      float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), _lib.ptr(self._step_dev), _lib.ptr(self._bc_dev), self.clip, _lib.stream())
        else:
            _lib.call("ssv_adamw_counted", a.numel, _lib.ptr(a.data), _lib.ptr(a.grad), _lib.ptr(g2), _lib.ptr(self.exp_avg), _lib.ptr(self.exp_avg_sq),
                      float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]), _lib.ptr(self._step_dev),
                      _lib.ptr(self._bc_dev), self.clip, _lib.stream())


def get_optimizer(config, params):
    name = config.get("name", "sgd")
    if name == "sgd":
        return FusedSGD(params, lr=config["lr"], weight_decay=config["weight_decay"], momentum=0.9)
    if name == "adamw":
        if config.get("amsgrad", False):
            raise NotImplementedError("AdamW with amsgrad=True is not built (configs/dino.yaml uses amsgrad: False)")
        return FusedAdamW(params, lr=config["lr"], weight_decay=config["weight_decay"], eps=config.get("epsilon", 1e-06))
    if name == "adam":
        raise NotImplementedError("optimizer adam (coupled weight decay) is not used by any accelerated algorithm; not built")
    raise NotImplementedError(f"Invalid optimizer {name}")


def get_scheduler(config, optimizer):
    name = config.get("name", None)
    warmup_epochs = config.get("warmup_epochs", 0)
    if warmup_epochs > 0:
        peak = optimizer.param_groups[0]["lr"]
        for group in optimizer.param_groups:
            group["lr"] = 1e-12 + peak / warmup_epochs
    if name is None:
        return None, warmup_epochs
    if name == "cosine":
        return lr_scheduler.CosineAnnealingLR(optimizer, config["epochs"] - warmup_epochs, eta_min=0.0, last_epoch=-1), warmup_epochs
    if name == "multistep":
        return lr_scheduler.MultiStepLR(optimizer, config["milestones"], config["gamma"]), warmup_epochs
    raise NotImplementedError(f"Invalid scheduler {name}")
