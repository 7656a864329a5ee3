"""Loss modules of the accelerated path with the reference's constructor signatures
(utils/losses.py:8-46 SimclrLoss, :120-142 BarlowLoss; BYOL uses nn.MSELoss, models/byol.py:89).

Each forward runs the fused HIP kernels (forward AND the gradient w.r.t. the embeddings), and
returns a 0-d tensor wired into torch.autograd so ``loss.backward()`` hands dz to the heads.
With a process group, the embeddings are all-gathered over RCCL so NT-Xent sees global negatives.
"""
import torch
import torch.nn as nn

from .. import _lib, ops
from .. import distributed as hdist


def _pad32(d):
    return (d + 31) // 32 * 32


class _NTXentFn(torch.autograd.Function):
    """NT-Xent over the GLOBAL batch.  Each rank normalises its rows straight into their slots of the gathered
    matrix Z = [zi_all ; zj_all], all-gathers the slots (RCCL), computes the log-sum-exp of ITS rows against all
    columns, and - because S is symmetric - needs only the all-gathered row LSEs to form the exact gradient of the
    global-mean loss w.r.t. its own rows (option iii): TWO small collectives per step - one all-gather of [zi ; zj], one of
    the row LSEs with the loss partial riding along - and no redundant Gram work."""

    @staticmethod
    def forward(ctx, zi, zj, normalize, temperature):
        b, d = zi.shape
        ld = _pad32(d)
        wide = ld > 128                  # beyond the register-resident kernels: Gram block through the GEMM kernels (any width)
        world, rank = hdist.world_size(), hdist.rank()
        nglob, seg0 = b * world, b * rank
        rows = (2 * nglob + 15) // 16 * 16 if wide else 2 * nglob          # wide: the row count is a GEMM dimension (multiple of 16)
        zall = torch.empty((rows, ld), dtype=torch.float32, device=zi.device)
        if rows > 2 * nglob:
            ops.fill_(zall[2 * nglob:], 0.0)
        exchange = hdist.is_on()          # also with a forced world of one rank (the single-GPU functional test of the RCCL calls)
        if exchange:
            # ONE all-gather for both views: every rank normalises into its [zi_r ; zj_r] block, the blocks are gathered in rank order and
            # re-laid as [zi_all ; zj_all] (the layout the row kernels index: row r < N pairs with row r + N)
            loc = torch.empty((2 * b, ld), dtype=torch.float32, device=zi.device)
            _, inv_i = ops.l2norm_fwd(zi.detach().contiguous(), normalize, ld, out=loc[:b])
            _, inv_j = ops.l2norm_fwd(zj.detach().contiguous(), normalize, ld, out=loc[b:])
            blocks = torch.empty((world * 2 * b, ld), dtype=torch.float32, device=zi.device)
            hdist.all_gather_blocks(blocks, loc)
            zall[:2 * nglob].view(2, world, b, ld).copy_(blocks.view(world, 2, b, ld).permute(1, 0, 2, 3))
        else:
            _, inv_i = ops.l2norm_fwd(zi.detach().contiguous(), normalize, ld, out=zall[seg0:seg0 + b])
            _, inv_j = ops.l2norm_fwd(zj.detach().contiguous(), normalize, ld, out=zall[nglob + seg0:nglob + seg0 + b])
        inv_t = 1.0 / float(temperature)
        gram = None
        if wide:
            zloc = torch.cat((zall[seg0:seg0 + b], zall[nglob + seg0:nglob + seg0 + b]), 0)
            gram = ops.conv2d_fwd(zloc.view(2 * b, 1, 1, ld), zall).view(2 * b, rows)          # S = Z_loc Z_all^T on the MFMA GEMM
            lse_loc, pos_loc = ops.ntxent_gram_fwd(gram, nglob, b, seg0, inv_t)
        else:
            lse_loc, pos_loc = ops.ntxent_fwd(zall, nglob, b, seg0, inv_t)
        loss = ops.ntxent_loss(lse_loc, pos_loc, 1.0 / (2 * nglob))
        if exchange:
            # ONE all-gather for the row log-sum-exps of both views; this rank's share of the loss rides along as one more float, so every
            # rank sums the same `world` partials in the same order (the global-batch loss, identical on all ranks, no all-reduce)
            pack = torch.empty((1, 2 * b + 4), dtype=torch.float32, device=zi.device)
            pack[0, :2 * b].copy_(lse_loc)
            pack[0, 2 * b:].copy_(loss.expand(4))
            got = torch.empty((world, 2 * b + 4), dtype=torch.float32, device=zi.device)
            hdist.all_gather_blocks(got, pack)
            loss = got[:, 2 * b].sum()
            lse_all = torch.empty((2 * nglob,), dtype=torch.float32, device=zi.device)
            lse_all.view(2, world, b).copy_(got[:, :2 * b].view(world, 2, b).permute(1, 0, 2))
        else:
            lse_all = lse_loc
        ctx.saved = (zall, lse_all, inv_i, inv_j, nglob, b, seg0, d, inv_t, bool(normalize), gram)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        zall, lse_all, inv_i, inv_j, nglob, b, seg0, d, inv_t, normalize, gram = ctx.saved
        if gram is not None:             # wide embeddings: W' in place over the Gram block, then dZ = W' Z_all as a GEMM
            ops.ntxent_gram_weights(gram, lse_all, nglob, b, seg0, inv_t, inv_t / (2 * nglob))
            dzall = ops.conv2d_dgrad(gram.view(2 * b, 1, 1, -1), zall, (2 * b, 1, 1, zall.shape[1])).view(2 * b, zall.shape[1])
        else:
            dzall = ops.ntxent_bwd(zall, lse_all, nglob, b, seg0, inv_t, inv_t / (2 * nglob))
        ops.scale_(dzall, dloss.contiguous())                # chain rule with the upstream scalar, read on the device
        dzi = ops.l2norm_bwd(zall[seg0:seg0 + b], inv_i, dzall[:b], d, normalize)
        dzj = ops.l2norm_bwd(zall[nglob + seg0:nglob + seg0 + b], inv_j, dzall[b:], d, normalize)
        return dzi, dzj, None, None


class SimclrLoss(nn.Module):
    """NT-Xent (utils/losses.py:8-46): defaults normalize=False, temperature=1.0 like the reference."""

    def __init__(self, normalize=False, temperature=1.0):
        super().__init__()
        self.normalize = normalize
        self.temperature = temperature

    def forward(self, zi, zj):
        if zi.shape != zj.shape or zi.dim() != 2:
            raise ValueError(f"SimclrLoss expects two [N,D] matrices, got {tuple(zi.shape)} and {tuple(zj.shape)}")
        return _NTXentFn.apply(zi, zj, self.normalize, self.temperature)


class _MSEPairFn(torch.autograd.Function):
    """Data parallel: the mean runs over the GLOBAL batch (scale 1/(B_local*world*D)) and the scalar is all-reduced, so the
    SUM of per-rank parameter gradients is the large-batch gradient - the same convention as the other two losses."""

    @staticmethod
    def forward(ctx, o1, o2, t1, t2):
        o1c, o2c, t1c, t2c = (t.detach().contiguous() for t in (o1, o2, t1, t2))
        loss, do1, do2 = ops.mse_pair(o1c, o2c, t1c, t2c, 1.0 / (o1c.numel() * hdist.world_size()))
        hdist.all_reduce_sum(loss)
        ctx.saved = (do1, do2)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        do1, do2 = ctx.saved
        g = dloss.contiguous()
        return ops.scale_(do1, g), ops.scale_(do2, g), None, None


def byol_pair_loss(online_1, online_2, target_1, target_2):
    """MSE(online_1, target_2) + MSE(online_2, target_1), each a mean over B*D (models/byol.py:129-130)."""
    return _MSEPairFn.apply(online_1, online_2, target_1, target_2)


class _BarlowFn(torch.autograd.Function):
    """Whole Barlow Twins loss + its gradient w.r.t. both embedding matrices, composed from the C ABI:
    [F.normalize] -> column standardisation (BN kernels, gamma = sqrt((B-1)/B), eps = 0 = unbiased std without eps)
    -> Craw = zi_hat^T zj_hat (wgrad-shaped MFMA GEMM over the batch) -> loss / G (ssv_barlow_cgrad)
    -> d zi_hat = zj_hat G^T (fwd-shaped GEMM), d zj_hat = zi_hat G (dgrad-shaped GEMM) -> BN backward [-> normalize backward].
    Data parallel: the embeddings are all-gathered and every rank evaluates the global-batch loss redundantly
    (3 GEMMs of B x D x D - cheaper than a 64 MB all-reduce of C), then keeps the gradient rows of its own samples."""

    @staticmethod
    def forward(ctx, zi, zj, normalize, lmbda):
        bl, d = zi.shape
        if d % 16:
            raise _lib.SsvError(f"BarlowLoss: projection dim must be a multiple of 16 (got {d})")
        world, rank = hdist.world_size(), hdist.rank()
        b = bl * world
        dev = zi.device
        zs = []
        for z in (zi, zj):
            zall = torch.empty((b, d), dtype=torch.float32, device=dev)
            zall[rank * bl:(rank + 1) * bl].copy_(z.detach())
            if world > 1:
                hdist.all_gather_rows(zall, bl)
            zs.append(zall)
        invs = [None, None]
        if normalize:
            for k in range(2):
                zs[k], invs[k] = ops.l2norm_fwd(zs[k], True)
        gamma = ops.fill_(torch.empty(d, dtype=torch.float32, device=dev), ((b - 1) / b) ** 0.5)
        beta = ops.fill_(torch.empty(d, dtype=torch.float32, device=dev), 0.0)
        hats, stats = [], []
        for k in range(2):
            y, mean, invstd = ops.bn_train_fwd(zs[k], gamma, beta, None, None, None, relu=False, eps=0.0)
            hats.append(y)
            stats.append((mean, invstd))
        craw = torch.empty((d, d), dtype=torch.float32, device=dev)
        ops.conv2d_wgrad(hats[1].view(b, 1, 1, d), hats[0].view(b, 1, 1, d), craw, craw, accumulate=False)   # craw[i][j] = sum_b zi_hat[b,i] zj_hat[b,j]
        loss, g = ops.barlow_cgrad(craw, 1.0 / b, lmbda)
        dhat_i = ops.conv2d_fwd(hats[1].view(b, 1, 1, d), g).view(b, d)                     # [b,i] = sum_j zj_hat[b,j] G[i,j]
        dhat_j = ops.conv2d_dgrad(hats[0].view(b, 1, 1, d), g, (b, 1, 1, d)).view(b, d)     # [b,j] = sum_i zi_hat[b,i] G[i,j]
        dgam, dbet = torch.empty(d, dtype=torch.float32, device=dev), torch.empty(d, dtype=torch.float32, device=dev)
        grads = []
        for k, dh in enumerate((dhat_i, dhat_j)):
            dz, _ = ops.bn_train_bwd(dh, None, zs[k], gamma, stats[k][0], stats[k][1], False, dgam, dbet, accumulate=False)
            if normalize:
                dz = ops.l2norm_bwd(zs[k], invs[k], dz, d, True)
            grads.append(dz[rank * bl:(rank + 1) * bl])
        ctx.saved = grads
        return loss

    @staticmethod
    def backward(ctx, dloss):
        dzi, dzj = ctx.saved
        g = dloss.contiguous()
        return ops.scale_(dzi.contiguous(), g), ops.scale_(dzj.contiguous(), g), None, None


class BarlowLoss(nn.Module):
    """Barlow Twins loss (utils/losses.py:120-142): defaults normalize=True, off_diagonal_weight=0.005 like the reference."""

    def __init__(self, normalize=True, off_diagonal_weight=0.005):
        super().__init__()
        self.normalize = normalize
        self.lmbda = off_diagonal_weight

    def forward(self, z_i, z_j):
        if z_i.shape != z_j.shape or z_i.dim() != 2:
            raise ValueError(f"BarlowLoss expects two [B,D] matrices, got {tuple(z_i.shape)} and {tuple(z_j.shape)}")
        return _BarlowFn.apply(z_i, z_j, self.normalize, self.lmbda)


# ------------------------------------------------------------------------------------------- sibling algorithms
class _NegDotPairFn(torch.autograd.Function):
    """0.5 * (SimSiamLoss(o1, t2) + SimSiamLoss(o2, t1)) (utils/losses.py:145-152, models/simsiam.py:126-127); the mean runs over
    the GLOBAL batch under data parallelism (weight 1/world, scalar all-reduced)."""

    @staticmethod
    def forward(ctx, o1, o2, t1, t2):
        o1c, o2c, t1c, t2c = (t.detach().contiguous() for t in (o1, o2, t1, t2))
        loss, do1, do2 = ops.negdot_pair(o1c, o2c, t1c, t2c, 0.5 / (o1c.shape[0] * hdist.world_size()))
        hdist.all_reduce_sum(loss)
        ctx.saved = (do1, do2)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        do1, do2 = ctx.saved
        g = dloss.contiguous()
        return ops.scale_(do1, g), ops.scale_(do2, g), None, None


def simsiam_pair_loss(online_1, online_2, target_1, target_2):
    return _NegDotPairFn.apply(online_1, online_2, target_1, target_2)


class _RelicKLFn(torch.autograd.Function):
    """alpha * KL term of RelicLoss (utils/losses.py:195-201) with its gradient w.r.t. all three embedding matrices."""

    @staticmethod
    def forward(ctx, zi, zj, zo, normalize, temperature, alpha):
        # Data parallel: the term soft-maxes the N diagonal logits ACROSS the batch, so the (normalised) embeddings are all-gathered and
        # every rank evaluates the global-batch term (N x D work - nothing next to the encoder), keeping the gradient rows of its own
        # samples: the SUM of the per-rank parameter gradients is the gradient of the global loss, like the other losses here.
        bl, d = zi.shape
        world, rank = hdist.world_size(), hdist.rank()
        mats, invs = [], []
        for z in (zi, zj, zo):
            zh, inv = ops.l2norm_fwd(z.detach().contiguous(), normalize)
            if world > 1:
                zall = torch.empty((bl * world, d), dtype=torch.float32, device=z.device)
                zall[rank * bl:(rank + 1) * bl].copy_(zh)
                zh = hdist.all_gather_rows(zall, bl)
            mats.append(zh)
            invs.append(inv)
        loss, *grads = ops.relic_kl(mats[0], mats[1], mats[2], 1.0 / float(temperature), alpha)
        mine = slice(rank * bl, (rank + 1) * bl)
        ctx.saved = tuple(ops.l2norm_bwd(mats[k][mine].contiguous(), invs[k], grads[k][mine].contiguous(), d, normalize) for k in range(3))
        return loss

    @staticmethod
    def backward(ctx, dloss):
        g = dloss.contiguous()
        dzi, dzj, dzo = (ops.scale_(t, g) for t in ctx.saved)
        return dzi, dzj, dzo, None, None, None


class RelicLoss(nn.Module):
    """utils/losses.py:155-201: the NT-Xent of (zi, zj) plus alpha * the invariance term against the un-augmented image's embedding."""

    def __init__(self, normalize=True, temperature=1.0, alpha=0.5):
        super().__init__()
        self.normalize, self.temperature, self.alpha = normalize, temperature, alpha
        self.contrastive = SimclrLoss(normalize, temperature)

    def forward(self, zi, zj, z_orig):
        return self.contrastive(zi, zj) + _RelicKLFn.apply(zi, zj, z_orig, self.normalize, self.temperature, self.alpha)


class _MocoFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, query, keys, bank, queue_size, normalize, temperature):
        # Data parallel: every query is scored against its own key and the (replicated) queue - a per-sample loss, so the mean over the
        # GLOBAL batch is the local mean weighted 1/world with the scalar all-reduced (the convention of the BYOL pair loss); the queue
        # stays identical on every rank because MemoryBank.add_batch pushes the all-gathered keys in rank order.
        n, d = query.shape
        world = hdist.world_size()
        qn, inv_q = ops.l2norm_fwd(query.detach().contiguous(), normalize)
        kn, _ = ops.l2norm_fwd(keys.detach().contiguous(), normalize)
        neg = ops.conv2d_fwd(qn.view(n, 1, 1, d), bank).view(n, bank.shape[0])                    # [N, K_pad] products with the queue (MFMA GEMM)
        loss, dq = ops.moco_loss(qn, kn, neg, queue_size, 1.0 / float(temperature))
        ops.conv2d_dgrad(neg.view(n, 1, 1, -1), bank, (n, 1, 1, d), addend=dq.view(n, 1, 1, d), out=dq.view(n, 1, 1, d))   # dq += P . bank
        dz = ops.l2norm_bwd(qn, inv_q, dq, d, normalize)
        if world > 1:
            w = torch.full((), 1.0 / world, dtype=torch.float32, device=query.device)
            ops.scale_(loss.view(1), w)
            ops.scale_(dz, w)
            hdist.all_reduce_sum(loss)
        ctx.saved = dz
        return loss

    @staticmethod
    def backward(ctx, dloss):
        return ops.scale_(ctx.saved, dloss.contiguous()), None, None, None, None, None


class MocoLoss(nn.Module):
    """utils/losses.py:49-71.  ``memory_vectors`` is the device queue [K_pad, D] (K_pad = queue size rounded up to 16, extra rows zero
    and masked out) as held by models.moco.MemoryBank."""

    def __init__(self, normalize=True, temperature=1.0):
        super().__init__()
        self.normalize, self.temperature = normalize, temperature

    def forward(self, query, keys, memory_vectors, queue_size=None):
        k = memory_vectors.shape[0] if queue_size is None else queue_size
        return _MocoFn.apply(query, keys, memory_vectors, k, self.normalize, self.temperature)
