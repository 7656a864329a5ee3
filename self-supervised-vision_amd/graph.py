"""The training step as ONE HIP graph (torch.cuda.CUDAGraph = hipGraph on ROCm), for the regime the reference actually ships: resnet18 on 32 x 32 images
(configs/*.yaml), where a step is ~500 kernel launches of a few microseconds each and the host's enqueue time (Python -> ctypes -> hipLaunchKernel, ~18 us per
launch) is as long as the GPU's work (tools/bench_cifar.py, profiles/r05_cifar_r18_*).  Replaying the captured step costs the host one hipGraphLaunch.

What is captured is the trainer's own ``train_step(batch)`` (models/simclr.py:86-95 and its siblings) - both view passes on their two streams, the loss, the
backward pass through torch.autograd, the fused optimizer update - called once under stream capture on STATIC input buffers; every later step copies its
batch into those buffers and replays.  Nothing of the step lives on the host:

  * the loss leaves through a device-to-host copy node into a pinned buffer (``nn.early_item`` under capture), read after the replay has finished;
  * BatchNorm running statistics / num_batches_tracked, the optimizer state and the gradient slabs are device memory that the kernels update in place;
  * the learning rate, weight decay and momentum are DEVICE memory too (round 6: ssv_sgd_nesterov_dev / ssv_adamw_counted_dev read them from four floats the
    optimizer owns; ``optim.push_hyper()`` rewrites those when a schedule has moved them, models/simclr.py:77-84), so ONE graph per input shape serves the whole run -
    rounds 4-5 baked them in as kernel arguments and re-captured (and destroyed) a graph every epoch;
  * scalars that still reach kernels as arguments (DINO's temperatures, MoCo's momentum: ``trainer.graph_key()``) key the graph: a bounded LRU of live graphs
    (MAX_LIVE per StepGraph) covers their schedules;
  * scratch (ops.workspace) used by captured kernels is allocated inside the capture, in the graph's private pool, so no later eager allocation can move it.

On small images the captured step also takes another kernel SELECTION than the eager one (ops.graph_dispatch): with no host cost per launch the Winograd forms pay
from 64 tiles and 64 channels, which is what turns batch 64 from 6.8 into 5.2 ms per step; ``graph_floors=False`` captures the eager selection (then the replayed
steps are the eager steps bit for bit - tests/test_gpu_graph.py checks both).

Not graphed (the call falls back to the eager step): a process group (collectives stay eager), trainers that keep per-step state on the host (``graph_safe``
False - none of the shipped trainers any more: MoCo's queue pointer lives in device memory), a batch whose shapes have no graph yet and differ from the common one only once (the ragged last batch of an epoch runs eagerly).
AdamW is graphable because its step count lives in device memory (``ssv_adamw_counted``); DINO's per-epoch scalars (temperatures, weight decay) are part of the key.
"""
import atexit
import gc
import os
import weakref
from collections import OrderedDict

import torch

from . import distributed as hdist
from . import nn as hnn
from . import ops

# SSV_STEP_GRAPH: "1" always (where possible), "0" never, "auto" (default) for small images only - where the step is launch-bound
MODE = os.environ.get("SSV_STEP_GRAPH", "auto")
AUTO_MAX_PIXELS = 64 * 64            # per image: CIFAR (32 x 32) and the like; at 224 x 224 the step is GPU-bound and the static input copies cost more than the launches
WARMUP_STEPS = 2                     # eager steps before a capture: the optimizer's first update (its first-step flag), allocator warm-up
MAX_LIVE = 4                         # captured graphs a StepGraph keeps (least recently replayed one retired first).  Each holds its private pool: the step's activations
                                     # and workspace - resnet18 at 32 x 32: 0.35 GB at batch 64, 2.6 GB at batch 512 (tools/exp/r06_graph_lru_stress.py prints it)
MAX_RETIRED = 16                     # retired graphs waiting for destruction, process-wide, before the oldest generation is destroyed

# ---- when a graph is DESTROYED -----------------------------------------------------------------------------------------------------------------------------------
# hipGraphExecDestroy racing the runtime's completion-handler thread for the graph's LAST launch corrupts the host heap (round 5: native backtrace in libhsa-runtime64's
# handler thread; never with the graphs kept alive).  Rounds 4-5 destroyed a graph right after its last replay - every epoch - behind a sleep.  Now nothing destroys a
# graph near its last launch: a graph that leaves service (LRU eviction, close(), its trainer's end) is RETIRED into a two-generation graveyard, and a generation is
# destroyed only after the device has gone idle TWICE with other work in between (_reap: at a later capture, or at interpreter exit while the HIP runtime is still
# up) - its last launch is then at least one whole capture older than the destruction.  No timing heuristic anywhere.
_LIVE = weakref.WeakSet()          # every StepGraph that may hold graphs
_RETIRED = [[], []]                # generation 0: retired since the last reap; generation 1: one reap old - the next reap destroys it


def _retire(recs):
    _RETIRED[0].extend(recs)


def _reap(final=False):
    """Device idle, no capture in progress: destroy the generation that has already survived one reap, age the other.  ``final`` (interpreter exit): both."""
    if not (_RETIRED[0] or _RETIRED[1]):
        return
    import torch as _t
    if _t.cuda.is_available():
        _t.cuda.synchronize()
    for _ in range(2 if final else 1):
        old, _RETIRED[1] = _RETIRED[1], _RETIRED[0]
        _RETIRED[0] = []
        for rec in old:
            try:
                rec[0].reset()             # CUDAGraph.reset = hipGraphExecDestroy + hipGraphDestroy; then the record's events, static inputs and workspace go
            except Exception:
                pass
        del old
        if final and _t.cuda.is_available():
            _t.cuda.synchronize()


def _close_all():
    """Graph objects must not be left to interpreter finalisation: a CUDAGraph destroyed after the HIP runtime's own teardown aborts the process (seen once as
    'Aborted (core dumped)' AFTER a green pytest run - the exit code of the whole session was lost)."""
    for sg in list(_LIVE):
        sg.close()
    _reap(final=True)


atexit.register(_close_all)


class StepGraph:
    """``sg = StepGraph(trainer); metrics = sg(batch)`` - the drop-in for ``trainer.train_step(batch)``.  ``weak=True`` (what TwoViewTrainer.step passes: the trainer
    holds its StepGraph) holds the trainer weakly: no reference cycle, so dropping the trainer destroys its graphs at once - by reference count, at a defined point -
    and not whenever the cycle collector runs."""

    def __init__(self, trainer, keys=None, mode=None, graph_floors=True, weak=False):
        self._trainer = weakref.ref(trainer) if weak else (lambda: trainer)
        self.keys = keys
        self.graph_floors = graph_floors   # small images: the captured step takes the Winograd forms from fewer tiles / channels (ops.graph_dispatch); False = the eager selection, bit for bit
        self.mode = MODE if mode is None else mode
        self.graphs = OrderedDict()  # key -> (graph, static inputs, pinned loss, (workspace buffers, event objects of the capture) kept alive); least recently replayed first
        self.eager_steps = 0
        self.seen = {}               # input-shape signature -> times met without a graph
        self.disabled = None         # reason, once capture has failed or the trainer is not graphable
        self.replays = 0
        self.captures = 0

    @property
    def trainer(self):
        t = self._trainer()
        if t is None:
            raise RuntimeError("StepGraph: the trainer it was built for no longer exists")
        return t

    def _drop(self, keys):
        """Take the graphs under `keys` out of service.  They are not destroyed here (see the note on destruction above): they retire, with everything recorded inside their
        capture, and a later reap destroys them."""
        recs = [self.graphs.pop(k) for k in keys if k in self.graphs]
        if recs:
            _retire(recs)

    def close(self):
        """Retire every captured graph (the StepGraph can capture again afterwards)."""
        self._drop(list(self.graphs))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- eligibility ---------------------------------------------------------------------------------------------------------
    def _tensors(self, batch):
        keys = self.keys or getattr(self.trainer, "graph_inputs", None) or [k for k, v in batch.items() if torch.is_tensor(v) and v.is_floating_point() and v.dim() >= 3]
        return {k: batch[k] for k in keys}

    def _why_not(self, ins):
        from .utils.train_utils import FusedAdamW, FusedSGD
        if self.mode == "0":
            return "SSV_STEP_GRAPH=0"
        if not getattr(self.trainer, "graph_safe", False):
            return f"{type(self.trainer).__name__}.train_step keeps per-step state on the host (graph_safe is False)"
        if hdist.is_on():
            return "a process group is active (collectives stay eager)"
        if not isinstance(getattr(self.trainer, "optim", None), (FusedSGD, FusedAdamW)):
            return "the optimizer is neither the fused SGD nor the fused AdamW"
        if not ins or not all(t.is_cuda for t in ins.values()):
            return "the batch is not on the GPU"
        if self.mode == "auto" and max(t.shape[-1] * t.shape[-2] for t in ins.values()) > AUTO_MAX_PIXELS:
            return "images larger than 64 x 64: the step is GPU-bound (SSV_STEP_GRAPH=1 forces the graph)"
        return None

    def _key(self, ins):
        """Everything a captured step bakes in: the input shapes and every scalar that reaches a kernel as an ARGUMENT - AdamW's betas / epsilon / clamp, plus whatever
        the trainer names in ``graph_key()`` (DINO: temperatures, centre momentum)."""
        g = self.trainer.optim.param_groups[0]
        opt = tuple(sorted((k, float(v) if isinstance(v, (int, float)) else tuple(float(x) for x in v)) for k, v in g.items()
                           if k in ("betas", "eps")))            # lr / weight decay / momentum are device memory (optim.push_hyper): not part of the key
        extra = tuple(self.trainer.graph_key()) if hasattr(self.trainer, "graph_key") else ()
        # ... and the arithmetic: it selects kernels AND (round 6) dispatch thresholds, so a step captured under one never replays under the other (ops.arithmetic())
        return (tuple((k, tuple(t.shape), t.stride()) for k, t in sorted(ins.items())), opt, float(getattr(self.trainer.optim, "clip", 0.0)), extra, ops.ARITHMETIC)

    # ---- the step ------------------------------------------------------------------------------------------------------------
    def __call__(self, batch):
        if self.disabled is not None:
            return self.trainer.train_step(batch)
        ins = self._tensors(batch)
        why = self._why_not(ins)
        if why is not None:
            if self.mode == "0" or not hdist.is_on():
                self.disabled = why          # permanent reasons; a process group may come and go in tests
            return self.trainer.train_step(batch)
        key = self._key(ins)
        rec = self.graphs.get(key)
        if rec is None:
            shape_sig = key[0]
            self.seen[shape_sig] = self.seen.get(shape_sig, 0) + 1
            # eager until the optimizer has made its first update; a shape met for the first time runs eagerly too (the ragged last batch of an epoch)
            if self.trainer.optim._steps < 1 or self.eager_steps < WARMUP_STEPS or (self.seen[shape_sig] < 2 and any(k[0] != shape_sig for k in self.graphs)):
                self.eager_steps += 1
                return self.trainer.train_step(batch)
            if self.captures >= 8 and self.replays < 4 * self.captures:
                # a scalar of the key moves nearly every step (a per-step learning-rate schedule): capturing costs more than launching - stay eager
                self.disabled = f"{self.captures} captures for {self.replays} replays: a scalar baked into the graph changes too often"
                self.close()
                return self.trainer.train_step(batch)
            if len(_RETIRED[0]) + len(_RETIRED[1]) >= MAX_RETIRED:
                _reap()                      # idle device, no capture in progress: the generation retired before the previous reap goes
            steps_before = self.trainer.optim._steps
            try:
                self.captures += 1
                rec = self._capture(batch, ins, key)
            except Exception as exc:         # never take the training run down: the eager step is always there
                self.disabled = f"capture failed: {type(exc).__name__}: {exc}"
                torch.cuda.synchronize()
                # the recorded (never executed) step filled the weight caches with filters that live in the discarded graph's pool, and counted an optimizer step
                ops.invalidate_weight_caches()
                self.trainer.optim._steps = steps_before
                return self.trainer.train_step(batch)
            self.trainer.optim._steps = steps_before          # recording a step executes nothing: the replays count
            while len(self.graphs) >= MAX_LIVE:
                self._drop([next(iter(self.graphs))])         # least recently replayed
            self.graphs[key] = rec
        self.graphs.move_to_end(key)
        graph, static, host, _ = rec
        for k, t in ins.items():
            static[k].copy_(t, non_blocking=True)
        self.trainer.optim.push_hyper()      # lr / weight decay / momentum as the schedules left them (a 16-byte copy when they moved)
        graph.replay()
        self.trainer.optim._steps += 1
        # the replay changed the weights without running a line of Python: whatever an EAGER forward between replays (feature extraction, a ragged batch) cached of
        # them - transposed filters, Winograd-transformed filters - is stale now (found by test_graph_survives_eager_work_between_replays: the second extraction used
        # the first one's transformed filters)
        ops.invalidate_weight_caches()
        torch.cuda.current_stream().synchronize()
        self.replays += 1
        return {"loss": float(host.item())}

    def _capture(self, batch, ins, key):
        static = {k: torch.empty_like(t) for k, t in ins.items()}
        for k, t in ins.items():
            static[k].copy_(t)
        host = torch.empty(1, dtype=torch.float32, pin_memory=True)
        sbatch = dict(batch)
        sbatch.update(static)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        eager_ws, ops.workspace.buf = ops.workspace.buf, {}
        prev = hnn.begin_capture(host)
        small = max(t.shape[-1] * t.shape[-2] for t in ins.values()) <= AUTO_MAX_PIXELS
        floors = ops.graph_dispatch() if (self.graph_floors and small) else None
        trainer = self.trainer
        gc_was_on = gc.isenabled()
        gc.disable()                       # no destructor of an unrelated object (another trainer's graphs, events) may run in the middle of a stream capture
        try:
            if floors is not None:
                floors.__enter__()
            with torch.cuda.graph(graph):
                trainer.train_step(sbatch)
        finally:
            if gc_was_on:
                gc.enable()
            if floors is not None:
                floors.__exit__(None, None, None)
            events = hnn.end_capture(prev)
            graph_ws, ops.workspace.buf = ops.workspace.buf, eager_ws
            ops.invalidate_weight_caches()     # whatever the recorded step cached of the weights (transposed / transformed / split filters) was never computed
        _LIVE.add(self)
        return graph, static, host, (graph_ws, events)

    def describe(self):
        return {"mode": self.mode, "graphs": len(self.graphs), "retired": len(_RETIRED[0]) + len(_RETIRED[1]), "captures": self.captures, "replays": self.replays,
                "eager_steps": self.eager_steps, "disabled": self.disabled}
