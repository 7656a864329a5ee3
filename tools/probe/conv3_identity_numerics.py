#!/usr/bin/env python3
"""Numerics of "conv3's backward without re-reading its output" (DESIGN 9): dW3 and the BatchNorm statistic from G = g^T a2, S = sum a2 and the Gram
matrix a2^T a2 instead of from c3 = a2 W3^T - fp32 against an fp64 evaluation of the direct form, plain and with a2 centred.  CPU, torch only."""
import torch
torch.manual_seed(0)
M, C, K = 16 * 56 * 56, 64, 256
a2 = torch.relu(torch.randn(M, C, dtype=torch.float64) * 1.0 + 0.3)
W = torch.randn(K, C, dtype=torch.float64) / C ** 0.5
g = torch.randn(M, K, dtype=torch.float64) * 1e-3


def direct(a2, W, g, dt):
    a2, W, g = a2.to(dt), W.to(dt), g.to(dt)
    c3 = a2 @ W.t()
    mean, var = c3.mean(0), c3.var(0, unbiased=False)
    inv = (var + 1e-5).rsqrt()
    xh = (c3 - mean) * inv
    sg, sgx = g.sum(0), (g * xh).sum(0)
    dy = inv * (g - sg / M - xh * sgx / M)            # gamma = 1
    return dy.t() @ a2, sgx


def identity(a2, W, g, dt, centre):
    a2, W, g = a2.to(dt), W.to(dt), g.to(dt)
    S = a2.sum(0)
    mu_a = S / M
    ac = a2 - mu_a if centre else a2
    gram = ac.t() @ ac                                 # (centred) Gram matrix
    G = g.t() @ ac                                     # g^T a2 (centred: g^T (a2 - mu))
    sg = g.sum(0)
    mean = W @ mu_a                                    # mean of c3 per output channel
    if centre:
        var = ((W @ gram) * W).sum(1) / M              # var_k = w_k^T Cov w_k
        sgc = (W * G).sum(1)                           # sum g (c3 - mean)
        Cc = W @ gram                                  # sum (c3 - mean) (a2 - mu)
        Gf = G + sg[:, None] * mu_a[None, :]           # g^T a2
    else:
        ex2 = ((W @ gram) * W).sum(1) / M
        var = ex2 - mean * mean
        sgc = (W * G).sum(1) - mean * sg
        Cc = W @ gram - mean[:, None] * S[None, :]     # sum (c3 - mean) a2
        Gf = G
    inv = (var + 1e-5).rsqrt()
    sgx = inv * sgc
    # dy = inv (g - sg/M - xh sgx/M);  dW = dy^T a2 = inv (G - sg S^T / M - inv sgx/M * sum (c3-mean) a2)
    dW = inv[:, None] * (Gf - sg[:, None] * S[None, :] / M - (inv * sgx / M)[:, None] * Cc)
    return dW, sgx


ref, ref_sgx = direct(a2, W, g, torch.float64)
rel = lambda x, r: float((x.double() - r).norm() / r.norm())
d32, s32 = direct(a2, W, g, torch.float32)
print(f"direct form, fp32                 : dW3 rel err {rel(d32, ref):.2e}   sum g xhat rel err {rel(s32, ref_sgx):.2e}")
for centre in (False, True):
    i64, _ = identity(a2, W, g, torch.float64, centre)
    i32, s = identity(a2, W, g, torch.float32, centre)
    print(f"identity, {'centred a2' if centre else 'plain a2  '} fp32 (fp64 {rel(i64, ref):.1e}): dW3 rel err {rel(i32, ref):.2e}   sum g xhat rel err {rel(s, ref_sgx):.2e}")
