"""Oracle (test infrastructure): loss restatements in plain torch fp32 on CPU.

  * l2_normalize   - F.normalize(p=2, dim=-1) as used at utils/losses.py:20-22, models/byol.py:47,59
  * ntxent_loss    - SimclrLoss.forward, utils/losses.py:15-46, in its Gram/log-sum-exp form
  * barlow_loss    - BarlowLoss.forward, utils/losses.py:127-142
  * byol_mse_loss  - nn.MSELoss() pair as used at models/byol.py:89,129-130
"""
import torch


def l2_normalize(z, eps=1e-12):
    """z / max(||z||_2, eps) row-wise."""
    return z / z.norm(p=2, dim=-1, keepdim=True).clamp_min(eps)


def ntxent_loss(zi, zj, normalize=False, temperature=1.0):
    """NT-Xent.  The reference builds four [N,N] Grams, gathers the positive (diag of ij / ji)
    into column 0 and the 2N-2 negatives behind it, then cross_entropy(label 0) averaged over
    2N rows (utils/losses.py:27-45).  Equivalent form used here: Z=[zi;zj], S=Z Z^T / tau,
    diagonal masked out, loss = mean_r( logsumexp_c S_rc - S_{r,pos(r)} ), pos(r) = r +- N."""
    n = zi.shape[0]
    if normalize:
        zi, zj = l2_normalize(zi), l2_normalize(zj)
    z = torch.cat([zi, zj], dim=0)
    s = (z @ z.t()) / temperature
    s = s.masked_fill(torch.eye(2 * n, dtype=torch.bool), float("-inf"))
    pos = torch.cat([torch.arange(n, 2 * n), torch.arange(0, n)])
    lse = torch.logsumexp(s, dim=1)
    return (lse - s[torch.arange(2 * n), pos]).mean()


def barlow_loss(zi, zj, normalize=True, off_diagonal_weight=0.005):
    """Standardise columns with the UNBIASED std and no eps (utils/losses.py:136-137),
    C = zi^T zj / B (:138), sum over W o (C - I)^2 with W = 1 on the diagonal and lambda off it."""
    if normalize:
        zi, zj = l2_normalize(zi), l2_normalize(zj)
    b, d = zi.shape
    zi = (zi - zi.mean(0)) / zi.std(0)
    zj = (zj - zj.mean(0)) / zj.std(0)
    c = (zi.t() @ zj) / b
    eye = torch.eye(d)
    w = torch.full((d, d), float(off_diagonal_weight)) * (1 - eye) + eye
    return (w * (c - eye) ** 2).sum()


def byol_mse_loss(online_1, online_2, target_1, target_2):
    """models/byol.py:129-130: MSE(o1,t2) + MSE(o2,t1), each a mean over B*D elements."""
    return ((online_1 - target_2) ** 2).mean() + ((online_2 - target_1) ** 2).mean()
