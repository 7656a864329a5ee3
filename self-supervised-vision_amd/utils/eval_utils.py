"""Evaluation helpers (outside the accelerated hot path; infrequent, host-side).

compute_neighbor_accuracy restates the reference's faiss IndexFlatIP 20-NN label agreement
(utils/eval_utils.py:13-21) with a chunked numpy inner-product + argpartition, so --task train does
not need faiss.  linear_evaluation is a small closed loop on frozen features.
"""
import numpy as np


def compute_neighbor_accuracy(fvecs, targets, k=20, chunk=2048):
    fvecs = np.ascontiguousarray(fvecs, dtype=np.float32)
    targets = np.asarray(targets)
    n = fvecs.shape[0]
    k = min(k, n - 1)
    agree = 0.0
    for s in range(0, n, chunk):
        sims = fvecs[s:s + chunk] @ fvecs.T                       # inner product, like IndexFlatIP
        idx = np.argpartition(-sims, k, axis=1)[:, :k + 1]        # k+1 best (includes the query itself)
        part = np.take_along_axis(sims, idx, axis=1)
        order = np.argsort(-part, axis=1, kind="stable")
        nbrs = np.take_along_axis(idx, order, axis=1)[:, 1:]      # drop the top hit, as the reference does
        agree += (targets[nbrs] == targets[s:s + chunk, None]).mean(axis=1).sum()
    return float(agree / n)


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
