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


def linear_evaluation(config, train_data, test_data, num_classes, device):
    """Linear probe on frozen (already L2-normalised) features: full-batch softmax regression in numpy.
    (The reference's version cannot run - SURVEY 2 row 11 - so only its intent is kept.)"""
    xtr, ytr = np.asarray(train_data["fvecs"], np.float64), np.asarray(train_data["labels"])
    xte, yte = np.asarray(test_data["fvecs"], np.float64), np.asarray(test_data["labels"])
    w = np.zeros((xtr.shape[1], num_classes))
    b = np.zeros(num_classes)
    onehot = np.eye(num_classes)[ytr]
    lr = float(config.get("lr", 0.1))
    for _ in range(int(config.get("epochs", 100))):
        logits = xtr @ w + b
        logits -= logits.max(1, keepdims=True)
        p = np.exp(logits)
        p /= p.sum(1, keepdims=True)
        g = (p - onehot) / len(xtr)
        w -= lr * 10 * (xtr.T @ g)
        b -= lr * 10 * g.sum(0)
    return float(((xte @ w + b).argmax(1) == yte).mean())
