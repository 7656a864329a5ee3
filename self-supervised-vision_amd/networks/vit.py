"""The reference's Transformer encoder (networks/vit.py) on the HIP path.

Constructor (`TransformerEncoder(config)`), parameter names / shapes / init draws and the numbers are the reference's,
including what makes it differ from a textbook ViT: positional embeddings are CONCATENATED to the raw patch pixels before the
single projection (:80-81, :102), q/k/v read the un-normalised input while the residual branch is LayerNorm(x) (:22-31), there
is no attention output projection, and the feed-forward block has the same `f(x) + LayerNorm(x)` shape (:43-46).

Execution: tokens live in one dense [B*T, hidden] matrix; every Linear is the fp32-MFMA implicit-GEMM kernel over those rows,
attention is the fused flash-style kernel pair of csrc/vit.hip (no T x T matrix in HBM), LayerNorm carries the residual
add, and the feed-forward's second GEMM adds the LayerNorm branch in its epilogue.  `return_attn=True` (visualisation only in
the reference) is refused: the probabilities are never materialised.
"""
import math

import torch
import torch.nn as nn

from .. import nn as hnn
from .. import ops


def _linear(din, dout, bias=True):
    """nn.Linear's init draws in nn.Linear's order (weight: kaiming_uniform a=sqrt 5; bias: U(+-1/sqrt fan_in))."""
    w = torch.empty(dout, din)
    nn.init.kaiming_uniform_(w, a=math.sqrt(5))
    if not bias:
        return hnn.HipLinear(din, dout, weight=w, bias=False)
    b = torch.empty(dout)
    bound = 1.0 / math.sqrt(din)
    nn.init.uniform_(b, -bound, bound)
    return hnn.HipLinear(din, dout, weight=w, bias=b)


class _Table(nn.Module):
    """nn.Embedding's state (weight ~ N(0,1)); looked up by position inside the token-assembly kernel."""

    def __init__(self, rows, dim):
        super().__init__()
        self.weight = nn.Parameter(nn.init.normal_(torch.empty(rows, dim)))


class EmbeddingLayer(nn.Module):
    def __init__(self, num_global_patches, num_local_patches, input_dim, embedding_dim):
        super().__init__()
        self.num_global_patches, self.num_local_patches = num_global_patches, num_local_patches
        self.cls_embedding = _Table(1, input_dim)
        self.pos_embedding_global = _Table(num_global_patches + 1, embedding_dim)
        self.pos_embedding_local = _Table(num_local_patches + 1, embedding_dim)

    def table_for(self, num_patches):
        if num_patches == self.num_global_patches:
            return self.pos_embedding_global.weight
        if num_patches == self.num_local_patches:
            return self.pos_embedding_local.weight
        raise RuntimeError(f"Num patches {num_patches} not matching global {self.num_global_patches} or local {self.num_local_patches} patches")


class MultiheadSelfAttention(hnn.HipModule):
    def __init__(self, hidden_dim, num_heads):
        super().__init__()
        if hidden_dim % num_heads or hidden_dim // num_heads != 64:
            raise NotImplementedError(f"the attention kernel is built for head size 64 (hidden {hidden_dim} / heads {num_heads})")
        self.heads, self.hidden_dim, self.head_size = num_heads, hidden_dim, hidden_dim // num_heads
        self.query, self.key, self.value = (_linear(hidden_dim, hidden_dim, bias=False) for _ in range(3))
        self.layer_norm = hnn.HipLayerNorm(hidden_dim)

    def _run(self, tape, x, batch, tokens):
        o = hnn.qkv_attention(tape, x, self.query.weight, self.key.weight, self.value.weight, batch, tokens, self.heads)
        return self.layer_norm._run(tape, x, addend=o)                       # attention(x) + LayerNorm(x)


class Feedforward(hnn.HipModule):
    def __init__(self, hidden_dim, intermediate_dim):
        super().__init__()
        self.fc1 = _linear(hidden_dim, intermediate_dim)
        self.fc2 = _linear(intermediate_dim, hidden_dim)
        self.layer_norm = hnn.HipLayerNorm(hidden_dim)

    def _run(self, tape, x):
        identity = self.layer_norm._run(tape, x)
        return hnn.ffn_gelu(tape, x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, addend=identity)   # GELU and residual in GEMM epilogues


class TransformerLayer(hnn.HipModule):
    def __init__(self, hidden_dim, intermediate_dim, num_attention_heads):
        super().__init__()
        self.attention = MultiheadSelfAttention(hidden_dim, num_attention_heads)
        self.feedfwd = Feedforward(hidden_dim, intermediate_dim)

    def _run(self, tape, x, batch, tokens):
        return self.feedfwd._run(tape, self.attention._run(tape, x, batch, tokens))


class TransformerEncoder(hnn.HipModule):
    """forward(img [B,3,H,W]) -> [B, hidden_dim]: the [CLS] embedding.  H/patch * W/patch must equal num_global_patches or
    num_local_patches."""

    def __init__(self, config):
        super().__init__()
        for key in ("hidden_dim", "embedding_dim", "intermediate_dim", "num_attention_heads", "patch_size", "num_encoder_layers",
                    "num_global_patches", "num_local_patches"):
            setattr(self, {"num_encoder_layers": "num_layers"}.get(key, key), config[key])
        pdim = 3 * self.patch_size ** 2
        if (pdim + self.embedding_dim) % 4 or self.hidden_dim % 16:
            raise NotImplementedError("3*patch^2 + embedding_dim must be a multiple of 4 and hidden_dim of 16 for the GEMM kernels")
        self.projection_fc = _linear(pdim + self.embedding_dim, self.hidden_dim)
        self.embedding = EmbeddingLayer(self.num_global_patches, self.num_local_patches, pdim, self.embedding_dim)
        self.enc_layers = nn.ModuleList([TransformerLayer(self.hidden_dim, self.intermediate_dim, self.num_attention_heads)
                                         for _ in range(self.num_layers)])
        self.out_dim = self.hidden_dim

    def _prepare_input(self, x):
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError(f"expected an image batch [B,3,H,W], got {tuple(x.shape)}")
        if x.shape[2] % self.patch_size or x.shape[3] % self.patch_size:
            raise RuntimeError(f"image size {tuple(x.shape[2:])} is not a multiple of the patch size {self.patch_size}")
        return ops.nchw_to_nhwc(x)

    def forward(self, img, return_attn=False):
        if return_attn:
            raise NotImplementedError("return_attn: the fused attention kernel never materialises the probabilities")
        return super().forward(img)

    _LAYERS_PER_STAGE = 3        # gradient buckets of ~20 MB for ViT-S (6.5 MB per layer): a few large collectives, each under the rest of the backward

    def grad_stages(self):
        """Gradient buckets of the data-parallel exchange (distributed.BucketedGradSync): token embedding + input projection, then the encoder layers in runs of
        three - contiguous parameter runs of the optimizer's arena in forward order, reduced last-run-first while the backward walks on (round 4 reduced the ViT's
        85 MB in one call at step())."""
        first = list(self.projection_fc.parameters()) + list(self.embedding.parameters())
        n = self._LAYERS_PER_STAGE
        return [first] + [[p for layer in self.enc_layers[i:i + n] for p in layer.parameters()] for i in range(0, len(self.enc_layers), n)]

    def _run(self, tape, x):
        batch = x.shape[0]
        num_patches = (x.shape[1] // self.patch_size) * (x.shape[2] // self.patch_size)
        pos = self.embedding.table_for(num_patches)
        hnn.stage_mark(tape, self, 0)
        tok, tokens = hnn.vit_embed(tape, x, self.embedding.cls_embedding.weight, pos, self.patch_size)
        h = self.projection_fc._run(tape, tok)
        for i, layer in enumerate(self.enc_layers):
            if i % self._LAYERS_PER_STAGE == 0:
                hnn.stage_mark(tape, self, 1 + i // self._LAYERS_PER_STAGE)      # fires in backward once this run's weight gradients are enqueued
            h = layer._run(tape, h, batch, tokens)
        return hnn.take_cls(tape, h, batch, tokens)
