"""TEST INFRASTRUCTURE - CPU restatement of the reference's kNN evaluation (utils/eval_utils.py:13-21).

The search itself lives in a third-party dependency absent from the reference tree and from this image: faiss (requirements.txt
lists `faiss-gpu`), `faiss.IndexFlatIP` = exhaustive search by inner product, each result row sorted by decreasing score.
Restated here in numpy; ties - which faiss leaves unspecified - go to the smaller index.  Pinned against
tests/golden/eval_level.npz, produced by the reference's own compute_neighbor_accuracy running over an exact-search stand-in for
the faiss index (tests/golden/gen_golden.py::eval_level).  Only tests/, smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np


def knn_indices(fvecs, k):
    """[n, k+1] neighbour ids by decreasing inner product (fp32 scores, stable order)."""
    x = np.ascontiguousarray(fvecs, dtype=np.float32)
    scores = x @ x.T
    return np.argsort(-scores, axis=1, kind="stable")[:, :k + 1]


def neighbor_agreement_count(fvecs, targets, k=20):
    """Number of (anchor, neighbour) pairs with equal labels over neighbours 1..k (the best hit is dropped, :16-18)."""
    targets = np.asarray(targets)
    nbrs = knn_indices(fvecs, k)[:, 1:]
    return int((targets[nbrs] == targets[:, None]).sum())


def compute_neighbor_accuracy(fvecs, targets, k=20):
    """utils/eval_utils.py:13-21: mean over the n x k table of label matches."""
    n = len(targets)
    return neighbor_agreement_count(fvecs, targets, k) / float(n * k)


def clustered_features(seed, n, d, classes, spread):
    """The seeded feature sets of tests/golden/eval_level.npz (same generator as gen_golden.clustered_features)."""
    import torch
    g = torch.Generator().manual_seed(seed)
    centres = torch.randn(classes, d, generator=g)
    labels = torch.randint(0, classes, (n,), generator=g)
    x = centres[labels] + spread * torch.randn(n, d, generator=g)
    return torch.nn.functional.normalize(x, dim=1).numpy(), labels.numpy()


def linear_evaluation(config, train_data, test_data, num_classes, batches):
    """CPU restatement (torch) of the linear probe the product runs: nn.Linear + NLLLoss(log_softmax), SGD(momentum, weight decay),
    cosine schedule per epoch - the intent of utils/eval_utils.py:37-76 (that function cannot run in the reference: "parity
    unpinned", the pin is this restatement).  `batches(n, batch_size, epoch, shuffle)` supplies the sample order."""
    import math
    import torch
    xtr, ytr = torch.as_tensor(train_data["fvecs"], dtype=torch.float32), torch.as_tensor(train_data["labels"], dtype=torch.long)
    xte, yte = torch.as_tensor(test_data["fvecs"], dtype=torch.float32), torch.as_tensor(test_data["labels"], dtype=torch.long)
    head = torch.nn.Linear(xtr.shape[1], num_classes)
    opt = torch.optim.SGD(head.parameters(), lr=config["lr"], momentum=config.get("momentum", 0.9), weight_decay=config.get("weight_decay", 1e-06))
    epochs, bs = int(config["epochs"]), int(config["batch_size"])
    accs = []
    for epoch in range(1, epochs + 1):
        for g in opt.param_groups:
            g["lr"] = 0.5 * config["lr"] * (1.0 + math.cos(math.pi * (epoch - 1) / epochs))
        for idx in batches(xtr.shape[0], bs, epoch, True):
            loss = torch.nn.functional.nll_loss(torch.log_softmax(head(xtr[idx]), -1), ytr[idx])
            opt.zero_grad()
            loss.backward()
            opt.step()
        if epoch == epochs:
            with torch.no_grad():
                accs = [float((head(xte[idx]).argmax(-1) == yte[idx]).float().mean()) for idx in batches(xte.shape[0], bs, epoch, False)]
    return float(np.mean(accs)), head
