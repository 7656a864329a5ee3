"""Oracle (test infrastructure): optimizer / schedule restatements.

  * sgd_nesterov_step - optim.SGD(momentum=0.9, nesterov=True, weight_decay=wd) as built by
                        get_optimizer, utils/train_utils.py:11-13 (YAML momentum/nesterov ignored)
  * seeded_lr         - get_scheduler's warm-up lr seeding, utils/train_utils.py:30-33
  * byol_tau          - BYOL.update_tau, models/byol.py:116-118
  * ema_update        - BYOL.momentum_update, models/byol.py:120-123
"""
import math

import torch


@torch.no_grad()
def sgd_nesterov_step(params, grads, bufs, lr, weight_decay, momentum=0.9):
    """In place.  g += wd*p; buf = g (first step) else momentum*buf + g; p -= lr*(g + momentum*buf).
    Weight decay hits every tensor incl. BN/bias.  ``bufs[i]`` is None before step 0."""
    for i, (p, g) in enumerate(zip(params, grads)):
        if g is None:
            continue
        g = g + weight_decay * p if weight_decay != 0 else g.clone()
        if bufs[i] is None:
            bufs[i] = g.clone()
        else:
            bufs[i].mul_(momentum).add_(g)
        p.sub_(lr * (g + momentum * bufs[i]))


def seeded_lr(lr, warmup_epochs):
    """lr in force during epoch 1: get_scheduler overwrites it with 1e-12 + lr/warmup_epochs."""
    return 1e-12 + lr / warmup_epochs if warmup_epochs > 0 else lr


def byol_tau(step, max_steps, tau_upper=1.0, tau_lower=0.996):
    return tau_upper - (tau_upper - tau_lower) * (math.cos(math.pi * step / max_steps) + 1) / 2


@torch.no_grad()
def ema_update(online_params, target_params, tau):
    """zip() truncates to the target's tensors: encoder + proj_head align, pred_head is dropped;
    buffers (BN running stats) are not touched."""
    for o, t in zip(online_params, target_params):
        t.copy_(tau * t + (1.0 - tau) * o)
