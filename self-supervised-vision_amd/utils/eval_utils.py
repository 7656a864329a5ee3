"""Evaluation helpers.

compute_neighbor_accuracy is the reference's 20-NN label agreement (utils/eval_utils.py:13-21, faiss.IndexFlatIP search) as
one C-ABI call: S = Z Z^T on the fp32-MFMA GEMM kernel, streaming top-(k+1) per query on the GPU (csrc/evalknn.hip) - no faiss,
and `eval_every` costs milliseconds.  linear_evaluation is a small closed loop on frozen features.
"""
import numpy as np
import torch

from .. import ops


def compute_neighbor_accuracy(fvecs, targets, k=20, device=None):
    """fvecs [n,d] and targets [n]: numpy arrays (what build_features returns) or tensors already on the GPU."""
    if not torch.cuda.is_available():
        raise RuntimeError("compute_neighbor_accuracy runs on the GPU (libssv_hip); no HIP device is visible and there is no CPU fallback")
    device = device or torch.device("cuda", torch.cuda.current_device())
    z = torch.as_tensor(fvecs, dtype=torch.float32).to(device)
    labels = torch.as_tensor(targets).to(device=device, dtype=torch.int32).contiguous()
    n = z.shape[0]
    if z.dim() != 2 or labels.shape != (n,):
        raise ValueError(f"expected fvecs [n,d] and targets [n], got {tuple(z.shape)} and {tuple(labels.shape)}")
    k = min(int(k), n - 1)                                          # tiny evaluation sets (synthetic smoke runs)
    return ops.knn_label_agreement(z, labels, k) / float(n * k)


def hungarian_match(preds, targets, preds_k, targets_k):
    from scipy.optimize import linear_sum_assignment
    votes = np.zeros((preds_k, targets_k))
    for c1 in range(preds_k):
        for c2 in range(targets_k):
            votes[c1, c2] = int(((preds == c1) & (targets == c2)).sum())
    rows, cols = linear_sum_assignment(preds.shape[0] - votes)
    return list(zip(rows.tolist(), cols.tolist()))


def probe_batches(n, batch_size, epoch, shuffle, seed=420):
    """Sample order of one probe epoch: a seeded permutation per epoch (the reference shuffles with DataLoader(shuffle=True))."""
    if not shuffle:
        order = torch.arange(n)
    else:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed + epoch))
    return [order[s:s + batch_size] for s in range(0, n, batch_size)]


def linear_evaluation(config, train_data, test_data, num_classes, device):
    """Linear probe on frozen features, on the GPU: nn.Linear -> NLLLoss(log_softmax) trained with SGD(momentum 0.9, weight decay
    1e-6) and a per-epoch cosine schedule, mini-batches of config["batch_size"]; returns the mean per-batch test accuracy of the
    last epoch.  That is the INTENT of utils/eval_utils.py:37-76 - the reference's own function cannot run (its loaders are tuples,
    it averages a bool tensor, and `input_dim` in the shipped configs does not match the feature width), so the width is taken
    from the features.  Logits and the weight gradient run on the MFMA GEMM kernels (classes padded to a multiple of 4)."""
    import math
    from .. import _lib
    if not torch.cuda.is_available():
        raise RuntimeError("linear_evaluation runs on the GPU (libssv_hip); no HIP device is visible and there is no CPU fallback")
    device = device if isinstance(device, torch.device) and device.type == "cuda" else torch.device("cuda", torch.cuda.current_device())
    to_dev = lambda a, dt: torch.as_tensor(np.asarray(a) if not torch.is_tensor(a) else a).to(device=device, dtype=dt).contiguous()
    xtr, ytr = to_dev(train_data["fvecs"], torch.float32), to_dev(train_data["labels"], torch.int32)
    xte, yte = to_dev(test_data["fvecs"], torch.float32), to_dev(test_data["labels"], torch.int32)
    for name, y in (("train", ytr), ("test", yte)):
        lo, hi = int(y.min().item()), int(y.max().item())
        if lo < 0 or hi >= num_classes:
            raise ValueError(f"linear_evaluation: {name} labels span [{lo}, {hi}] but num_classes = {num_classes}")
    d = xtr.shape[1]
    if d % 4:
        pad = 4 - d % 4
        xtr, xte = torch.nn.functional.pad(xtr, (0, pad)), torch.nn.functional.pad(xte, (0, pad))
        d += pad
    cpad = (num_classes + 3) // 4 * 4
    head = torch.nn.Linear(train_data["fvecs"].shape[1], num_classes)           # the reference's init draw (CPU RNG)
    flat = torch.zeros(cpad * d + cpad, device=device)                          # [W (cpad x d) | b (cpad)] in one buffer: one optimiser launch
    w, b = flat[:cpad * d].view(cpad, d), flat[cpad * d:]
    w[:num_classes, :head.in_features].copy_(head.weight.detach())
    b[:num_classes].copy_(head.bias.detach())
    grad, buf = torch.zeros_like(flat), torch.zeros_like(flat)
    gw, gb = grad[:cpad * d].view(cpad, d), grad[cpad * d:]
    stats = torch.zeros(2, device=device)
    lr0, epochs, bs = float(config["lr"]), int(config["epochs"]), int(config["batch_size"])
    mom, wd = float(config.get("momentum", 0.9)), float(config.get("weight_decay", 1e-06))
    lib = _lib.load()

    def run(x, y, train, lr, first):
        n = x.shape[0]
        logits = ops.conv2d_fwd(x.view(n, 1, 1, d), w, bias=b).view(n, cpad)
        dlog = torch.empty_like(logits) if train else None
        ws = _lib.workspace.get(lib.ssv_softmax_ce_workspace_bytes(n), device)
        _lib.call("ssv_softmax_ce_fwd_bwd", n, num_classes, cpad, _lib.ptr(logits), _lib.ptr(y), _lib.ptr(stats), _lib.ptr(dlog), _lib.ptr(ws), ws.numel(), _lib.stream())
        if train:
            ops.conv2d_wgrad(x.view(n, 1, 1, d), dlog.view(n, 1, 1, cpad), w, gw, accumulate=False)
            ops.colsum(dlog, gb, accumulate=False)
            _lib.call("ssv_sgd", flat.numel(), _lib.ptr(flat), _lib.ptr(grad), _lib.ptr(buf), lr, wd, mom, 0, int(first), _lib.stream())
            ops.invalidate_weight_caches()         # w changed under its cached bf16 planes (every in-place update of a GEMM operand says so)
        return stats

    acc, step = 0.0, 0
    for epoch in range(1, epochs + 1):
        lr = 0.5 * lr0 * (1.0 + math.cos(math.pi * (epoch - 1) / epochs))      # CosineAnnealingLR(T_max=epochs), stepped once per epoch
        for idx in probe_batches(xtr.shape[0], bs, epoch, True):
            idx = idx.to(device)
            run(xtr[idx], ytr[idx], True, lr, step == 0)
            step += 1
        if epoch == epochs:
            accs = [float(run(xte[idx.to(device)], yte[idx.to(device)], False, 0.0, False)[1]) for idx in probe_batches(xte.shape[0], bs, epoch, False)]
            acc = float(np.mean(accs))
    print("\nCompleted linear evaluation. Average validation accuracy is {:.2f}%".format(100 * acc))
    return acc
