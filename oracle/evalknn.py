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
