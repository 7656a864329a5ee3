"""MoCo on the HIP path - drop-in for the reference trainer (models/moco.py:23-126).

Kept from the reference: the key encoder starts as a copy of the query encoder and follows it by EMA (m = 0.999) after every
optimiser step; the queue starts as zeros and receives the L2-normalised keys AFTER that update; the head is one Linear layer
behind nn.ReLU (which is the identity on ResNet's pooled features - an average of post-ReLU activations - in value and in
gradient, so no kernel is spent on it).  The queue lives in HBM ([K rounded up to 16, D], extra rows zero and masked) and is
pushed by one kernel instead of a per-row Python loop (models/moco.py:32-37)."""
import torch

from .. import _lib, nn as hnn, ops
from ..utils import losses, train_utils
from .base import NETWORKS, TwoViewTrainer
from .heads import _fresh_linear


class MemoryBank:
    def __init__(self, queue_size, feature_size, device):
        self.size = int(queue_size)
        self.bank = ops.fill_(torch.empty(((self.size + 15) // 16 * 16, feature_size), dtype=torch.float32, device=device), 0.0)
        # the write pointer lives in device memory (ssv_queue_push_counted reads and advances it): a replayed step (graph.StepGraph) moves it like an eager one
        self._ptr_dev = torch.zeros(1, dtype=torch.int32, device=device)

    @property
    def ptr(self):
        return int(self._ptr_dev.item())

    def add_batch(self, batch):
        """Data parallel: the keys of ALL ranks enter every rank's queue, in rank order - the replicas stay identical."""
        from .. import distributed as hdist
        keys = batch.detach().contiguous()
        if hdist.is_on():
            b, world = keys.shape[0], hdist.world_size()
            allk = torch.empty((b * world, keys.shape[1]), dtype=keys.dtype, device=keys.device)
            allk[hdist.rank() * b:(hdist.rank() + 1) * b].copy_(keys)
            keys = hdist.all_gather_rows(allk, b)
        ops.queue_push_counted(self.bank, self.size, self._ptr_dev, keys)

    def get_vectors(self):
        return self.bank


class EncoderModel(hnn.HipModule):
    def __init__(self, encoder, encoder_dim, projection_dim):
        super().__init__()
        self.encoder = encoder
        self.proj_head = _fresh_linear(encoder_dim, projection_dim)

    def _prepare_input(self, x):
        return self.encoder._prepare_input(x)

    def _run(self, tape, x):
        return self.proj_head._run(tape, self.encoder._run(tape, x))     # ReLU(pooled features) == pooled features


class MoCo(TwoViewTrainer):
    algo = "moco"
    graph_safe = True    # the queue pointer is device memory (MemoryBank); momentum update and queue push are part of the step and of its graph
    graph_inputs = ("aug_1", "aug_2")

    def _build(self, arch):
        encoder, encoder_dim = NETWORKS[arch].values()
        cfg = self.config
        self.query_encoder = EncoderModel(encoder(**cfg["encoder"]), encoder_dim, cfg["proj_dim"]).to(self.device)
        self.key_encoder = EncoderModel(encoder(**cfg["encoder"]), encoder_dim, cfg["proj_dim"]).to(self.device)   # its init draws are consumed, then overwritten
        self.memory_bank = MemoryBank(cfg["queue_size"], cfg["proj_dim"], self.device)
        self.m = cfg.get("momentum", 0.999)
        self.key_encoder.load_state_dict(self.query_encoder.state_dict())
        for p in self.key_encoder.parameters():
            p.requires_grad = False
        self.optim = train_utils.get_optimizer(cfg["optimizer"], params=self.query_encoder.parameters())
        self._key_arena = train_utils.ParamArena(list(self.key_encoder.parameters()), with_grads=False)
        self.loss_fn = losses.MocoLoss(**cfg["loss_fn"])

    @torch.no_grad()
    def momentum_update(self):
        ops.ema_(self._key_arena.data, self.optim.arena.data, self.m)

    def _embed(self, img):
        return self.query_encoder(img)

    def graph_key(self):
        return (float(self.m),)

    def train_step(self, batch):
        img_1, img_2 = batch["aug_1"].to(self.device), batch["aug_2"].to(self.device)
        with hnn.parallel_views(self.device) as pv:
            with pv.view(0):
                query = self.query_encoder(img_1)
            with pv.view(1):
                with torch.no_grad():
                    keys = self.key_encoder(img_2)
        loss = self.loss_fn(query, keys, self.memory_bank.get_vectors(), self.memory_bank.size)
        loss_now = hnn.early_item(loss)                  # the scalar leaves for the host now; the backward does not wait for it, nor it for the backward
        self.optim.zero_grad()
        loss.backward()
        self.optim.step()
        self.momentum_update()
        self.memory_bank.add_batch(keys)
        return {"loss": loss_now.get()}

    def _checkpoint_state(self):
        return {"encoder": self.query_encoder.state_dict()}

    def _load_state(self, state):
        self.query_encoder.load_state_dict(state["encoder"])
