#!/usr/bin/env python3
"""CPU numerics of Winograd F(4x4, 3x3) in fp32 against an fp64 direct convolution, next to F(2x2, 3x3) and the direct fp32 convolution
(torch CPU): forward, data gradient (= forward with the rotated filter), weight gradient.  Interpolation points {0, +-1, +-2, inf} (Lavin &
Gray) and the {0, +-1, +-1/2, inf} set (smaller transform entries).  Decides whether an F(4x4) GPU probe is worth building (go/no-go bar:
error vs fp64 <= 3x the direct kernel's).      python tools/probe/wino44_numerics.py"""
import torch
import torch.nn.functional as F

torch.manual_seed(0)


def mats(points):
    """Cook-Toom matrices for F(m, 3) from finite interpolation points (+ infinity): returns AT [m x a], G [a x 3], BT [a x a] in fp64."""
    import numpy as np
    from numpy.polynomial import polynomial as P
    a = len(points) + 1
    m = a - 2
    pts = np.array(points, dtype=np.float64)
    # Vandermonde forms (Barabasz et al. notation): AT[i][j] = p_j^i, last column = [0..0,1]; G[j][k] = p_j^k / N_j; BT from Lagrange basis polynomials
    AT = np.zeros((m, a)); G = np.zeros((a, 3)); BT = np.zeros((a, a))
    for j, p in enumerate(pts):
        N = np.prod([p - q for k, q in enumerate(pts) if k != j])
        for i in range(m):
            AT[i, j] = p ** i
        for k in range(3):
            G[j, k] = p ** k / N
        # coefficients of prod_{k != j} (x - p_k)
        poly = np.array([1.0])
        for k, q in enumerate(pts):
            if k != j:
                poly = P.polymul(poly, np.array([-q, 1.0]))
        BT[j, :len(poly)] = poly
    AT[m - 1, a - 1] = 1.0
    G[a - 1, 2] = 1.0
    full = np.array([1.0])
    for q in pts:
        full = P.polymul(full, np.array([-q, 1.0]))
    BT[a - 1, :len(full)] = full
    return torch.tensor(AT), torch.tensor(G), torch.tensor(BT)


def check(AT, G, BT):
    m, a = AT.shape
    d, g = torch.randn(a, dtype=torch.float64), torch.randn(3, dtype=torch.float64)
    y = AT @ ((G @ g) * (BT @ d))
    ref = torch.stack([sum(d[i + k] * g[k] for k in range(3)) for i in range(m)])
    assert torch.allclose(y, ref, atol=1e-10), (y, ref)


def wino_fwd(x, w, AT, G, BT, dt):
    """x [N,C,H,W] (H, W multiples of m after padding by 1), w [K,C,3,3]; everything computed in dtype dt."""
    m, a = AT.shape
    AT, G, BT = AT.to(dt), G.to(dt), BT.to(dt)
    n, c, h, wd = x.shape
    th, tw = -(-h // m), -(-wd // m)
    xp = F.pad(x.to(dt), (1, 1 + tw * m - wd, 1, 1 + th * m - h))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)                     # [N,C,th,tw,a,a]
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT)        # BT d B
    U = torch.einsum("ij,kcjl,ml->kcim", G, w.to(dt), G)           # G g GT  [K,C,a,a]
    M = torch.einsum("nctuil,kcil->nktuil", V, U)
    Y = torch.einsum("ij,nktujl,ml->nktuim", AT, M, AT)            # [N,K,th,tw,m,m]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(n, w.shape[0], th * m, tw * m)
    return y[:, :, :h, :wd]


def wino_wgrad(x, dy, AT, G, BT, dt):
    """dW [K,C,3,3] = sum over tiles of GT-side transform: dU = sum (BT d B) * (A dy AT) ; dW = GT dU G."""
    m, a = AT.shape
    AT, G, BT = AT.to(dt), G.to(dt), BT.to(dt)
    n, c, h, wd = x.shape
    th, tw = -(-h // m), -(-wd // m)
    xp = F.pad(x.to(dt), (1, 1 + tw * m - wd, 1, 1 + th * m - h))
    tiles = xp.unfold(2, a, m).unfold(3, a, m)
    V = torch.einsum("ij,nctujk,lk->nctuil", BT, tiles, BT)
    dyp = F.pad(dy.to(dt), (0, tw * m - wd, 0, th * m - h))
    dyt = dyp.unfold(2, m, m).unfold(3, m, m)                      # [N,K,th,tw,m,m]
    dM = torch.einsum("ji,nktujl,lm->nktuim", AT, dyt, AT)         # A dy AT: [a x a]
    dU = torch.einsum("nctuil,nktuil->kcil", V, dM)
    return torch.einsum("ji,kcjl,lm->kcim", G, dU, G)              # GT dU G


def rel(a, b):
    return float((a.double() - b).norm() / b.norm())


if __name__ == "__main__":
    sets = {"F(2x2) {0,1,-1}": mats([0, 1, -1]), "F(4x4) {0,1,-1,2,-2}": mats([0, 1, -1, 2, -2]), "F(4x4) {0,1,-1,1/2,-1/2}": mats([0, 1, -1, 0.5, -0.5]),
            "F(4x4) {0,1,-1,1/2,-2}": mats([0, 1, -1, 0.5, -2])}
    for v in sets.values():
        check(*v)
    for name, n, h, c in (("p128.1.conv2", 8, 28, 128), ("p256.1.conv2", 16, 14, 256), ("p512.1.conv2", 32, 7, 512)):
        x = torch.relu(torch.randn(n, c, h, h) * 1.0 + 0.1)       # a post-BatchNorm-ReLU activation
        w = torch.randn(c, c, 3, 3) * (2.0 / (9 * c)) ** 0.5
        dy = torch.randn(n, c, h, h)
        y64 = F.conv2d(x.double(), w.double(), padding=1)
        dw64 = torch.nn.grad.conv2d_weight(x.double(), w.shape, dy.double(), padding=1)
        line = f"{name}: direct fp32 (torch CPU) fwd {rel(F.conv2d(x, w, padding=1), y64):.2e} wgrad {rel(torch.nn.grad.conv2d_weight(x, w.shape, dy, padding=1), dw64):.2e}"
        print(line)
        for tag, (AT, G, BT) in sets.items():
            assert rel(wino_fwd(x, w, AT, G, BT, torch.float64), y64) < 1e-12
            assert rel(wino_wgrad(x, dy, AT, G, BT, torch.float64), dw64) < 1e-12
            print(f"    {tag:28s} fp32: fwd {rel(wino_fwd(x, w, AT, G, BT, torch.float32), y64):.2e}   wgrad {rel(wino_wgrad(x, dy, AT, G, BT, torch.float32), dw64):.2e}")
