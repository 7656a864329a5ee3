"""Host-side layer engine: a small reverse-mode tape over the HIP ops, bridged into torch.autograd
at MODULE granularity (one autograd node per network call), so the reference's
``loss.backward(); optim.step()`` surface keeps working while every kernel on the path is ours.

Parameters are ordinary ``nn.Parameter``s (state_dict keys/shapes identical to the reference);
parameter gradients are accumulated by the kernels straight into ``p.grad`` (which the optimizer
keeps as views of one flat arena), never through torch ops.
"""
import os

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import SsvError


# ------------------------------------------------------------------------------------------- two-view concurrency
# The two views of a step are independent until the loss: their forward (and backward) passes are enqueued on two
# HIP streams so that the MFMA-bound convolutions of one view overlap the HBM-bound BatchNorm kernels - and the tail
# waves - of the other.  What has to be kept apart:
#   * scratch: _lib.workspace is per stream;
#   * parameter gradients: view slot 1 accumulates into a second gradient slab (p._grad_alt), the optimizer adds the
#     slabs in fixed order - bitwise the same result as one slab, no atomics;
#   * BatchNorm running statistics: slot 1's BN waits (HIP event) for the same layer's BN of slot 0, so the update
#     order is view 1 then view 2, exactly like the reference's sequential forward passes.
_SLOT = 0
_STREAMS = {}
_VIEW_STREAMS = os.environ.get("SSV_SINGLE_STREAM", "0") != "1"
# (round 6: the view-skew and stream-priority switches of round 4's experiments e4 / e8 are gone - both measured no effect, profiles/r04_experiments_step_time.txt)


def view_streams():
    return _VIEW_STREAMS


def set_view_streams(on):
    """Process-wide default of parallel_views: True = one HIP stream per view, False = both views on the ambient stream
    (what a per-kernel profile needs).  Returns the previous setting."""
    global _VIEW_STREAMS
    prev, _VIEW_STREAMS = _VIEW_STREAMS, bool(on)
    return prev


def current_slot():
    return _SLOT


def _dev_key(device):
    """One key per physical device: 'cuda', 'cuda:0' and torch.device('cuda', 0) must not get separate stream pairs."""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None and torch.cuda.is_available():
        device = torch.device("cuda", torch.cuda.current_device())
    return device


def join_view_streams(device):
    """Make the ambient stream wait for both view streams (kernels enqueued by Function.backward on them)."""
    device = _dev_key(device)
    if device in _STREAMS:
        cur = torch.cuda.current_stream(device)
        for st in _STREAMS[device]:
            if st != cur:
                _wait_stream(cur, st)


def _wait_stream(waiter, st):
    """``waiter.wait_stream(st)`` with an event object of our own: under stream capture it must outlive the capture (kept with the graph, _CAPTURE_KEEP)."""
    if _CAPTURE_KEEP is None:
        waiter.wait_stream(st)
        return
    ev = torch.cuda.Event()
    ev.record(st)
    waiter.wait_event(ev)
    _CAPTURE_KEEP.append(ev)


class parallel_views:
    """with parallel_views(device) as pv:  with pv.view(0): z1 = net(x1);  with pv.view(1): z2 = net(x2)
    On exit the ambient stream waits for both view streams.  Backward needs nothing special: torch.autograd runs
    each node on the stream of its forward and joins the streams at the end of backward()."""

    def __init__(self, device, enabled=None):
        device = _dev_key(device)
        self.device, self.enabled = device, (_VIEW_STREAMS if enabled is None else enabled) and device.type == "cuda"

    def __enter__(self):
        if self.enabled:
            _view_stream_pair(self.device)
            self.main = torch.cuda.current_stream(self.device)
            self.start = torch.cuda.Event()
            self.start.record(self.main)
            if _CAPTURE_KEEP is not None:
                _CAPTURE_KEEP.append(self.start)
            self.used = set()
        return self

    def view(self, slot):
        return _ViewCtx(self, slot)

    def __exit__(self, *exc):
        if self.enabled:
            for slot in self.used:
                _wait_stream(self.main, _STREAMS[self.device][slot])
        return False


class _ViewCtx:
    def __init__(self, pv, slot):
        self.pv, self.slot = pv, slot

    def __enter__(self):
        global _SLOT
        self.prev = _SLOT
        if self.pv.enabled:
            st = _STREAMS[self.pv.device][self.slot]
            st.wait_event(self.pv.start)
            self.ctx = torch.cuda.stream(st)
            self.ctx.__enter__()
            self.pv.used.add(self.slot)
            _SLOT = self.slot
        return self

    def __exit__(self, *exc):
        global _SLOT
        if self.pv.enabled:
            self.ctx.__exit__(*exc)
        _SLOT = self.prev
        return False


_INPUT_STREAM = os.environ.get("SSV_NO_INPUT_STREAM", "0") != "1"     # diagnostic switch: the next batch's augmentation on the ambient stream
_INPUT_STREAMS = {}
_DATA_READY = {}       # device -> event recorded behind the latest producer of resident input data (data_ready)


def data_ready(device, stream=None):
    """Producers of data that ``input_stream`` blocks will read - a loader's upload of its dataset, a source tensor generated on the device -
    call this once the producing work is ENQUEUED (on ``stream``, default: the ambient stream): every later ``input_stream`` block waits for it.
    Without it only the very first block of a process is ordered behind the ambient stream, and a second loader or a dataset uploaded with
    ``non_blocking`` could be read before it is complete."""
    device = _dev_key(device)
    if device.type != "cuda":
        return
    ev = torch.cuda.Event()
    ev.record(stream if stream is not None else torch.cuda.current_stream(device))
    _DATA_READY[device] = ev


def _view_stream_pair(device):
    device = _dev_key(device)
    if device not in _STREAMS:
        _STREAMS[device] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _STREAMS[device]


class input_stream:
    """with input_stream(device) as ins:  batch = <augmentation kernels>;  ins.publish(*tensors)
    The kernels inside run on a stream of their own, so the NEXT batch's views are built while the previous step's backward is still
    executing (the ambient stream is ordered behind that backward by the optimizer's join; with ``early_item`` the host gets here long before
    it has run).  Everything they read must be resident data whose producer called ``data_ready`` (every entry waits for the latest such event; the
    first entry also waits for the ambient stream once) or tensors created inside the block.  ``publish`` tells the caching allocator which other streams will read the outputs; on exit the ambient
    stream waits for the block's kernels.  Disabled (or on the CPU) the block simply runs on the ambient stream."""

    def __init__(self, device):
        self.device = _dev_key(device)
        self.enabled = _INPUT_STREAM and _VIEW_STREAMS and self.device.type == "cuda"

    def __enter__(self):
        if self.enabled:
            self.main = torch.cuda.current_stream(self.device)
            side = _INPUT_STREAMS.get(self.device)
            if side is None:
                side = _INPUT_STREAMS[self.device] = torch.cuda.Stream(self.device)
                side.wait_stream(self.main)               # whatever built the dataset on the ambient stream
            ready = _DATA_READY.get(self.device)
            if ready is not None:
                side.wait_event(ready)                    # EVERY entry: the latest producer of resident data (a no-op once it has completed)
            self.side = side
            self.ctx = torch.cuda.stream(side)
            self.ctx.__enter__()
        return self

    def publish(self, *tensors):
        if self.enabled:
            readers = (self.main,) + tuple(_view_stream_pair(self.device))
            for t in tensors:
                if torch.is_tensor(t) and t.is_cuda:
                    for st in readers:
                        t.record_stream(st)

    def __exit__(self, *exc):
        if self.enabled:
            done = torch.cuda.Event()
            done.record(self.side)
            self.ctx.__exit__(*exc)
            self.main.wait_event(done)
        return False


_EARLY_LOSS = os.environ.get("SSV_LATE_LOSS_READ", "0") != "1"      # diagnostic switch: read the loss with .item() after the update (drains the queue)
_CAPTURE_HOST = None     # while graph.StepGraph captures a step: the pinned float the step's loss is copied to by a node of the graph


_CAPTURE_ORDER = None    # while a step is captured: id(BatchNorm module) -> the event view 0 recorded behind its update IN THIS CAPTURE (see _bn_order_record)
_CAPTURE_KEEP = None     # while a step is captured: every event object recorded in the capture; graph.StepGraph keeps the list for as long as the graph lives


def begin_capture(host):
    """The step that follows is recorded into a HIP graph, not executed: ``early_item`` must neither synchronise nor allocate - it copies the scalar into
    ``host`` (pinned, owned by the graph's record) and ``get()`` returns a placeholder; the replaying caller reads ``host`` after the graph has run.
    The BatchNorm ordering events of the captured step are capture-local (``_CAPTURE_ORDER``): no event object that outlives the capture is ever recorded in it."""
    global _CAPTURE_HOST, _CAPTURE_ORDER, _CAPTURE_KEEP
    prev, _CAPTURE_HOST = (_CAPTURE_HOST, _CAPTURE_ORDER, _CAPTURE_KEEP), host
    _CAPTURE_ORDER = {}
    _CAPTURE_KEEP = []
    return prev


def capturing():
    """Is a training step being recorded into a HIP graph right now (graph.StepGraph._capture)?"""
    return _CAPTURE_HOST is not None


def end_capture(prev):
    global _CAPTURE_HOST, _CAPTURE_ORDER, _CAPTURE_KEEP
    kept = _CAPTURE_KEEP
    _CAPTURE_HOST, _CAPTURE_ORDER, _CAPTURE_KEEP = prev
    return kept                                   # the capture's event objects: the caller keeps them until the graph is gone


class early_item:
    """``loss.item()`` that does not drain the queue (reference: ``return {"loss": loss.item()}`` after ``optim.step()``, models/simclr.py:95).
    Constructed right after the loss kernel, BEFORE the backward is enqueued: the scalar travels to pinned host memory on the ambient stream and
    ``get()`` waits for that copy only.  ``train_step`` then returns while the backward and the update are still executing, and the host enqueues
    the next step's augmentation and first view behind them - otherwise every step starts with one view stream running alone for as long as
    the host needs to enqueue the first view's forward (12 ms of a 228 ms step at bs 512, and the other view finishes that much later)."""

    def __init__(self, t):
        self.t = t.detach()
        self.ev = None
        self.captured = _CAPTURE_HOST is not None and self.t.is_cuda
        if self.captured:
            _CAPTURE_HOST.copy_(self.t.reshape(1), non_blocking=True)          # a device-to-host copy node of the graph being captured
            return
        if self.t.is_cuda and _EARLY_LOSS:
            self.host = torch.empty(1, dtype=self.t.dtype, pin_memory=True)
            self.host.copy_(self.t.reshape(1), non_blocking=True)
            self.ev = torch.cuda.Event()
            self.ev.record(torch.cuda.current_stream(self.t.device))

    def get(self):
        if self.captured:
            return float("nan")              # nothing has executed yet: the value exists after the replay (graph.StepGraph reads the pinned buffer)
        if self.ev is None:
            return self.t.item()
        self.ev.synchronize()
        return self.host.item()


# ------------------------------------------------------------------------------------------- tape
_MARK = object()


def stage_mark(tape, module, stage):
    """Data parallel (distributed.BucketedGradSync): tell the gradient exchange, during backward, that every parameter gradient of
    ``module``'s stage ``stage`` has been enqueued by this pass.  Call it BEFORE running the stage's forward ops."""
    sync = None if tape is None else tape.sync
    if sync is not None and sync.has(module, stage):
        expected = getattr(tape, "expected", None)
        tape.mark(lambda: sync.ready(module, stage, expected))


class Tape:
    """Records (inputs, output, backward closure) per op.  ``backward`` walks it in reverse; a
    closure gets the gradient of its output plus any gradient already accumulated for each of
    its inputs, so producers that can fuse ``+=`` (conv dgrad's addend) do so."""

    def __init__(self, root, root_needs_grad):
        self.ops = []
        self.root = root
        self.root_needs_grad = root_needs_grad
        self.slot = _SLOT                     # which gradient slab this pass accumulates into
        self.sync = None                      # data parallel: the gradient exchange that wants to hear when a bucket is complete
        self.expected = None                  # ... and the buckets this pass was counted for (BucketedGradSync.expect)
        self.uses = {}                        # id(input) -> number of recorded ops that read it
        self.last = ()                        # during backward: per input of the running op, "no other op will add to its gradient"

    def record(self, inputs, output, bwd):
        self.ops.append((inputs, output, bwd))
        for t in inputs:
            if t is not None:
                self.uses[id(t)] = self.uses.get(id(t), 0) + 1

    def mark(self, fn):
        """``fn()`` is called during backward once every op recorded AFTER this point has run its backward closure - i.e. when the parameter
        gradients of everything that follows in the forward order have been enqueued (data parallel: a gradient bucket is complete)."""
        self.ops.append((_MARK, None, fn))

    def needs_grad(self, t):
        return t is not self.root or self.root_needs_grad

    def backward(self, out, dout):
        grads = {id(out): dout}
        uses = self.uses
        while self.ops:
            inputs, output, bwd = self.ops.pop()
            if inputs is _MARK:
                bwd()
                continue
            g = grads.pop(id(output), None)
            for t in inputs:
                if t is not None:
                    uses[id(t)] -= 1
            if g is None:
                continue
            existing = [None if t is None else grads.get(id(t)) for t in inputs]
            self.last = tuple(t is not None and uses[id(t)] == 0 for t in inputs)
            new = bwd(g, existing)
            for t, ng in zip(inputs, new):
                if t is not None and ng is not None:
                    grads[id(t)] = ng
        return grads.get(id(self.root))


def _accum(existing, fresh):
    """Generic fallback when a producer cannot fuse the accumulation."""
    if existing is None:
        return fresh
    if isinstance(existing, ops.StridedGrad):          # a compact stride-2 gradient met a consumer that cannot take it as its addend
        existing = existing.materialize()
    return ops.add_(existing, fresh)


def grad_of(p, slot=0):
    """The kernels accumulate into p.grad (view slot 0) or p._grad_alt (slot 1, when the optimizer provides the
    second slab); created zeroed on first use (layout = p's layout)."""
    if slot == 1:
        alt = getattr(p, "_grad_alt", None)
        if alt is not None:
            return alt
        raise SsvError("a pass in view slot 1 needs the optimizer's second gradient slab (build the optimizer with "
                       "train_utils.get_optimizer before running parallel_views)")
    if p.grad is None:
        p.grad = ops.fill_(torch.empty_like(p), 0.0)
    return p.grad


# ------------------------------------------------------------------------------------------- ops on the tape
_FUSE_BN_STATS = os.environ.get("SSV_NO_BN_STATS_FUSION", "0") != "1"
_PAD_STEM = os.environ.get("SSV_NO_STEM_PADDING", "0") != "1"


_FUSE_BN_APPLY = os.environ.get("SSV_NO_BN_APPLY_FUSION", "0") != "1"      # diagnostic switch: materialise every activation
_FUSE_BN_BWD = os.environ.get("SSV_NO_BN_BWD_FUSION", "0") != "1"          # diagnostic switch: BatchNorm backward with its own reduction pass
_FUSE_BN_DY = os.environ.get("SSV_NO_BN_DY_FUSION", "0") != "1"            # diagnostic switch: BatchNorm backward always writes dx
# ... for feature maps of at least this many pixels.  Forming dx on load costs the consumers a second operand stream and ~3 VALU per element
# between their barriers; it pays where the removed pass is long (56x56 / 28x28 maps: 257.3 -> 254.6 ms per step at bs 512) and not on the
# small deep maps (all layers: 256.9 ms) - measured with SSV_BN_DY_MIN_HW = 0 / 784 / 3136 / off, three runs each.
# Round 6: on the bf16x3 arithmetic the consumers are bound by bytes, not by the matrix pipe, and the 14x14 maps pay too (re-measured with the traffic counters,
# profiles/r06_probe_thresholds_traffic.txt: both thresholds at 196 +0.4 % images/s, 648 -> 640 GB per step, BatchNorm passes 20 -> 14 ms).  Default by
# arithmetic: 196 on bf16x3, 784 on fp32 MFMA; SSV_BN_DY_MIN_HW=<pixels> overrides both (tests assign the module variable).
_BN_DY_MIN_HW = int(os.environ["SSV_BN_DY_MIN_HW"]) if os.environ.get("SSV_BN_DY_MIN_HW") else None


def _bn_dy_min_hw():
    return _BN_DY_MIN_HW if _BN_DY_MIN_HW is not None else (196 if ops.ARITHMETIC == "bf16x3" else 784)


_BN_DY_MIN_K = int(os.environ.get("SSV_BN_DY_MIN_K", "0"))                 # diagnostic: only convolutions with at least this many output channels
_FUSE_CLOSING = os.environ.get("SSV_NO_CLOSING_FUSION", "0") != "1"        # diagnostic switch: the closing activation of a unit gets its own pass
# ... on feature maps of [lo, hi] pixels.  Measured at bs 512 (three runs each, profiles/r02_experiments_step_time.txt exp12): off 252.8 ms, every
# stage 250.3, the 56x56 / 28x28 stages only 249.8, the 14x14 / 7x7 stages only 252.4 - as for the BatchNorm-backward operand, the pass is
# worth removing where it is long.
# (round 6, bf16x3: from 196 pixels - see _BN_DY_MIN_HW above)
_CLOSING_HW = tuple(int(v) for v in os.environ["SSV_CLOSING_HW"].split(",")) if os.environ.get("SSV_CLOSING_HW") else None


def _closing_hw():
    return _CLOSING_HW if _CLOSING_HW is not None else ((196 if ops.ARITHMETIC == "bf16x3" else 784), 10 ** 9)


_FUSE_SHORTCUT_GATE = os.environ.get("SSV_NO_SHORTCUT_GATE", "0") != "1"   # diagnostic switch: the projection shortcut's BatchNorm backward reduces in its own pass
_FUSE_BN_APPLY_3X3 = os.environ.get("SSV_NO_BN_APPLY_FUSION_3X3", "0") != "1"   # diagnostic switch: fuse the input BatchNorm of 1x1 convolutions only
_FUSE_NARROW_WINO = os.environ.get("SSV_NO_NARROW_WINO_INPUT_FUSION", "0") != "1"   # diagnostic switch: narrow (< 128 channels) Winograd layers get a materialised input
_COMPACT_S2_DGRAD = os.environ.get("SSV_NO_COMPACT_S2_DGRAD", "0") != "1"        # diagnostic switch: the stride-2 shortcut's data gradient at full resolution


class LazyAct:
    """relu?(raw * scale[c] + shift[c]): the output of conv -> BatchNorm(-> ReLU) that is NOT written to HBM.  Its consumers form it
    on the fly: a convolution while it stages its input (forward and weight gradient), the closing BatchNorm of a residual unit when it
    adds the projection shortcut.  ``raw`` is the producer's conv output (kept for the backward anyway); scale / shift / mean / invstd
    come from ssv_bn_stats_finalize.  On the tape it stands where the activation tensor would: gradients are keyed by this object."""
    __slots__ = ("raw", "scale", "shift", "mean", "invstd", "relu", "_bn_gate")

    def __init__(self, raw, scale, shift, mean, invstd, relu):
        self.raw, self.scale, self.shift, self.mean, self.invstd, self.relu = raw, scale, shift, mean, invstd, relu
        self._bn_gate = None

    @property
    def shape(self):
        return self.raw.shape

    def materialize(self):
        return ops.bn_apply(self.raw, self.scale, self.shift, relu=self.relu)[0]


class LazySum:
    """a = relu(raw * scale + shift + shortcut): the closing activation of a residual unit (networks/resnet.py:73-74), NOT YET written.  Its
    first consumer decides: the 1x1 conv1 of the next unit forms it while it stages its input and writes it out of the same kernel
    (ops.conv2d_fwd_sumin - the element-wise pass disappears); anything else (`tensor()`) runs that pass.  Either way the tensor exists
    afterwards (`t`): the next residual add, the weight gradient and the backward's ReLU mask need it.  On the tape the object stands where
    the tensor would: gradients are keyed by it."""
    __slots__ = ("raw", "scale", "shift", "mean", "invstd", "res", "res_affine", "want_mask", "second", "t", "mask", "_bn_gate")

    def __init__(self, raw, scale, shift, mean, invstd, res, res_affine, want_mask, second=None):
        self.raw, self.scale, self.shift, self.mean, self.invstd = raw, scale, shift, mean, invstd
        self.res, self.res_affine, self.want_mask, self.second = res, res_affine, want_mask, second
        self.t, self.mask, self._bn_gate = None, None, None

    @property
    def shape(self):
        return self.raw.shape

    def set(self, t, mask):
        self.t, self.mask = t, mask
        if mask is not None:
            self._bn_gate = ops.BnGateCtx(self.raw, self.mean, self.invstd, mask=mask, second=self.second)

    def tensor(self):
        if self.t is None:
            self.set(*ops.bn_apply(self.raw, self.scale, self.shift, relu=True, residual=self.res, res_affine=self.res_affine, want_mask=self.want_mask))
        return self.t


def _tensor(x):
    return x.tensor() if isinstance(x, LazySum) else x


def conv(tape, x, weight, stride, pad, bias=None, bn_stats=False, compact_dx=False):
    """``compact_dx`` (a 1x1 / stride-2 projection shortcut whose input's other consumer is a wide 1x1 conv1): when this is the FIRST contribution
    to the input's gradient it is handed on compact (ops.StridedGrad: one dense GEMM on the subsampled grid) and conv1's data gradient adds it
    in its epilogue - the full-resolution tensor of three quarters zeros is never written nor re-read.
    ``bn_stats``: the caller normalises the output next - let the conv epilogue produce the statistics partials (kept on the
    output tensor as ``_bn_partials`` for `batchnorm`), which saves BatchNorm's own pass over the conv output.
    ``x`` may be a LazyAct (conv -> BN -> ReLU output that was never written): the kernels then apply it while staging."""
    lazy = x if isinstance(x, LazyAct) else None
    want = bn_stats and bias is None and _FUSE_BN_STATS
    if lazy is not None and not (lazy.relu and bias is None and ops.can_fuse_conv_input(weight.shape[1], weight.shape[0])):
        raise SsvError("a lazy activation reached a convolution that cannot fuse it (the producer must check ops.can_fuse_conv_input)")
    affine = None if lazy is None else (lazy.scale, lazy.shift)
    y = None
    if isinstance(x, LazySum):
        if (x.t is None and want and _FUSE_CLOSING and ops.can_form_closing_sum(weight.shape, stride, pad)
                and _closing_hw()[0] <= x.shape[1] * x.shape[2] <= _closing_hw()[1]):
            y, part, a, mask = ops.conv2d_fwd_sumin(x.raw, x.res, x.scale, x.shift, x.res_affine, weight, want_mask=x.want_mask)
            x.set(a, mask)
            y._bn_partials = part
        src = x.tensor()
    else:
        src = x if lazy is None else lazy.raw
    if y is not None:
        pass
    elif lazy is not None:
        y, part = ops.conv2d_fwd_fused(src, weight, stride, pad, in_affine=affine, want_stats=want, keep_v=tape is not None and _WINO_KEEP_V)
        if part is not None:
            y._bn_partials = part
    else:
        fused = ops.conv2d_fwd_stats(src, weight, stride, pad, keep_v=tape is not None and _WINO_KEEP_V) if want else None
        if fused is not None:
            y = fused[0]
            y._bn_partials = tuple(fused[1:])
        else:
            y = ops.conv2d_fwd(src, weight, stride, pad, bias=bias)
    if (tape is not None and bias is None and _FUSE_BN_DY and ops.can_lazy_dy(weight.shape, stride, pad) and y.shape[1] * y.shape[2] >= _bn_dy_min_hw()
            and weight.shape[0] >= _BN_DY_MIN_K):
        y._lazy_dy_ok = True       # a BatchNorm behind this output may hand its backward over as an ops.LazyGrad (formed by wgrad / dgrad)
    if (tape is not None and lazy is not None and bias is None and _FUSE_BN_DY and _FUSE_BN_BWD
            and ops.wino44_lazy_dy_ok(y, tuple(weight.shape), src.shape)):
        y._lazy_dy_ok = True       # all three products of this layer run Winograd F(4x4): its dY transform forms the BatchNorm backward on load (ops.WINOGRAD44_DY_BOTH)
    if tape is not None:
        need_dx = lazy is not None or tape.needs_grad(x)

        slot = tape.slot
        wino_v = y.__dict__.pop("_wino_v", None)       # Winograd forward: its transformed input is the weight gradient's operand

        def bwd(dy, existing):
            # x is the output of a BatchNorm + ReLU and nothing else will add to its gradient: gate + reduce in the data gradient's epilogue
            gate = getattr(x, "_bn_gate", None) if (need_dx and tape.last[0] and _FUSE_BN_BWD) else None
            # will the data gradient take the plain Winograd path (ops.conv2d_dgrad: no addend, no second gate target)?  Only then does the Winograd dY transform
            # of the weight gradient also write the data gradient's transformed operand (2.25x dY) - otherwise it would be written for nothing and live as long as dy
            plain_dgrad = need_dx and existing[0] is None and (gate is None or getattr(gate, "x2", None) is None)
            ops.conv2d_wgrad(src, dy, weight, grad_of(weight, slot), stride, pad, accumulate=True, in_affine=affine, wino_v=wino_v,
                             dbias=None if bias is None else grad_of(bias, slot), dgrad_follows=plain_dgrad)
            if not need_dx:
                return (None,)
            ex = existing[0]
            if (compact_dx and _COMPACT_S2_DGRAD and _FUSE_BN_BWD and ex is None and not tape.last[0] and stride == 2 and pad == 0
                    and weight.shape[2] == 1 and weight.shape[3] == 1 and not isinstance(dy, ops.LazyGrad)):
                return (ops.compact_s2_dgrad(dy, weight, src.shape),)
            dx = ops.conv2d_dgrad(dy, weight, src.shape, stride, pad, addend=ex, out=ex, gate=gate)
            return (dx,)
        tape.record((x,), y, bwd)
    return y


# Memory switch: the Winograd forward keeps its transformed input V (16 x tiles x C floats = 4x the convolution's input, 5.2x on 7x7 maps: ~10 GB per
# step at bs 512 over the ten Winograd layers and two views) for the weight gradient.  SSV_WINOGRAD_KEEP_V=0 drops it: the weight gradient of those
# layers then runs on the direct implicit-GEMM kernel (about 2x slower per layer), everything else unchanged.
_WINO_KEEP_V = os.environ.get("SSV_WINOGRAD_KEEP_V", "1") != "0"
_ROW_STEM = os.environ.get("SSV_NO_ROW_STEM", "0") != "1"          # diagnostic switch: the 3-channel stem with both operands padded to 4 channels


def stem_conv(tape, x, weight, stride, pad, bn_stats=False):
    """The 3-channel image convolution (networks/resnet.py:96-99, 147).  No input gradient (images).  ``bn_stats``: the epilogue leaves the
    BatchNorm statistics partials, as in `conv`.

    Row-taps form (default): the image stays unpadded; for one filter row the S taps x 3 channels of an output pixel are 3 S contiguous floats
    of the NHWC image row, so the contraction runs over R rows of 24 floats (the filter's rows zero-padded from 3 S) - 168 columns for the 7x7
    stem's 147 real ones.  (Round 1-2 form, `SSV_NO_ROW_STEM=1`: both operands zero-padded to 4 channels, one filter tap per 16-byte load:
    224 columns, and a padding pass over the images.)"""
    k, c, r, s_ = weight.shape
    want = bn_stats and _FUSE_BN_STATS
    if _ROW_STEM and ops.can_row_stem(weight.shape):
        wrows = ops.stem_weight_rows(weight)
        y, part = ops.stem_conv_fwd(x, wrows, tuple(weight.shape), stride, pad, want_stats=want)
        if part is not None:
            y._bn_partials = part
        if tape is not None:
            slot = tape.slot

            def bwd_rows(dy, existing):
                dwr = ops.stem_conv_wgrad(x, dy, tuple(weight.shape), stride, pad)                       # [K*R][24]
                ops.unpad_channels(dwr, grad_of(weight, slot).permute(0, 2, 3, 1).reshape(k * r, s_ * c), accumulate=True)
                return (None,)
            tape.record((x,), y, bwd_rows)
        return y
    xp = ops.pad_channels(x, 4)                                                  # [N,H,W,4]
    wp = ops.pad_channels(weight.permute(0, 2, 3, 1), 4).permute(0, 3, 1, 2)     # OHWI [K,R,S,4], seen as [K,4,R,S] channels_last
    fused = ops.conv2d_fwd_stats(xp, wp, stride, pad) if want else None
    if fused is not None:
        y = fused[0]
        y._bn_partials = tuple(fused[1:])
    else:
        y = ops.conv2d_fwd(xp, wp, stride, pad)
    if tape is not None:
        slot = tape.slot

        def bwd(dy, existing):
            dwp = torch.empty_like(wp)
            ops.conv2d_wgrad(xp, dy, wp, dwp, stride, pad, accumulate=False)
            ops.unpad_channels(dwp.permute(0, 2, 3, 1), grad_of(weight, slot).permute(0, 2, 3, 1), accumulate=True)
            return (None,)
        tape.record((x,), y, bwd)
    return y


_GROUP_AWARE = os.environ.get("SSV_NO_GROUP_AWARE_TILES", "0") != "1"     # diagnostic switch: grouped convolutions as plain dense block-diagonal ones


def grouped_conv(tape, x, weight, groups, stride, pad):
    """conv2d(groups=g): the grouped filter bank is expanded to its dense block-diagonal form (exact: the zeros contribute 0) and
    run on the MFMA kernels, each output-column tile contracting only over the channels of the groups it falls into (width / 64 times fewer
    k-tiles than the dense product); the dense-layout weight gradient's diagonal blocks are gathered back into the grouped parameter's gradient."""
    wd = ops.group_expand(weight, groups)
    g = groups if _GROUP_AWARE else 1          # > 1: every tile contracts over the channels of its own groups only (exact: the rest is zeros)
    y = ops.conv2d_fwd(x, wd, stride, pad, groups=g)
    if tape is not None:
        need_dx = tape.needs_grad(x)
        slot = tape.slot

        def bwd(dy, existing):
            dwd = torch.empty_like(wd)
            ops.conv2d_wgrad(x, dy, wd, dwd, stride, pad, accumulate=False, groups=g)
            ops.group_extract(dwd, grad_of(weight, slot), groups, accumulate=True)
            if not need_dx:
                return (None,)
            ex = existing[0]
            return (ops.conv2d_dgrad(dy, wd, x.shape, stride, pad, addend=ex, out=ex, groups=g),)
        tape.record((x,), y, bwd)
    return y


def _bn_order_wait(bn, x):
    if _STREAMS and _SLOT == 1:                                     # running-stat update order across the two view streams: slot 0 first
        ev = bn._order_event if _CAPTURE_ORDER is None else _CAPTURE_ORDER.get(id(bn))
        if ev is not None:
            torch.cuda.current_stream(x.device).wait_event(ev)


def _bn_order_record(bn, x):
    if _STREAMS and _SLOT == 0 and _lib.stream() != 0:              # on a view stream (the default stream's raw handle is 0): no Stream object per BatchNorm call
        if _CAPTURE_ORDER is not None:
            # Under stream capture a FRESH event per record, kept with the graph (_CAPTURE_KEEP): the module's persistent event is never recorded inside a capture
            # (an event recorded there belongs to that graph; it is neither re-recorded nor waited on eagerly afterwards).
            ev = torch.cuda.Event()
            ev.record()
            _CAPTURE_ORDER[id(bn)] = ev
            _CAPTURE_KEEP.append(ev)
            return
        if bn._order_event is None:
            object.__setattr__(bn, "_order_event", torch.cuda.Event())
        bn._order_event.record()                                   # on the current stream


def batchnorm(tape, x, bn, relu=False, residual=None, lazy=False, defer=False):
    """BatchNorm (batch statistics) [+ residual] [+ ReLU] of a conv output.

    ``defer=True`` (the closing BatchNorm + shortcut + ReLU of a residual unit): only the statistics are finalised here; the result is a
    LazySum whose first consumer either forms and writes it (conv1 of the next unit) or runs the element-wise pass.

    ``lazy=True`` (the caller guarantees that the only consumer is a convolution that can fuse its input, or - without ReLU - the
    closing BatchNorm of a residual unit): nothing but the statistics is computed; the result is a LazyAct.  ``residual`` may be such a
    LazyAct (the projection shortcut's BatchNorm folded into this kernel)."""
    partials = x.__dict__.pop("_bn_partials", None)
    lazy = lazy and partials is not None and residual is None and _FUSE_BN_APPLY
    res_lazy = residual if isinstance(residual, LazyAct) else None
    if res_lazy is not None and (partials is None or res_lazy.relu):
        raise SsvError("a lazy residual needs the statistics partials of this BatchNorm's input and must not carry a ReLU")
    res_t = None if res_lazy is not None else _tensor(residual)          # a LazySum shortcut is needed as a tensor here
    defer = defer and relu and residual is not None and partials is not None and _FUSE_CLOSING and _FUSE_BN_APPLY
    hold = second = None
    _bn_order_wait(bn, x)
    if lazy or res_lazy is not None or defer:
        m, c = ops._rows(x)
        mean, invstd, scale, shift = ops.bn_stats_finalize(m, c, partials, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                           eps=bn.eps, momentum=bn.momentum)
        _bn_order_record(bn, x)
        if lazy:
            out = LazyAct(x, scale, shift, mean, invstd, relu)
            if tape is not None:
                slot = tape.slot

                if relu:
                    out._bn_gate = ops.BnGateCtx(x, mean, invstd, scale=scale, shift=shift)

                def bwd_lazy(dact, existing):
                    part = dact.__dict__.pop("_gate_partials", None)      # consumed here: the buffer may be reused for another gradient
                    if part is not None and existing[0] is None and tape.last[0] and getattr(x, "_lazy_dy_ok", False):
                        coef = ops.bn_bwd_coef(x, bn.weight, mean, invstd, part, grad_of(bn.weight, slot), grad_of(bn.bias, slot))
                        return (ops.LazyGrad(dact, x, coef),)      # dx is formed by the producing convolution's wgrad / dgrad
                    if part is not None:          # the consumer's data gradient already gated dact and reduced it
                        dx = ops.bn_bwd_from_partials(dact, x, bn.weight, mean, invstd, part, grad_of(bn.weight, slot), grad_of(bn.bias, slot))
                    elif relu:
                        dx = ops.bn_relu_bwd_affine(dact, x, bn.weight, mean, invstd, scale, shift, grad_of(bn.weight, slot), grad_of(bn.bias, slot))
                    else:
                        dx, _ = ops.bn_train_bwd(dact, None, x, bn.weight, mean, invstd, False, grad_of(bn.weight, slot), grad_of(bn.bias, slot), accumulate=True)
                    return (_accum(existing[0], dx),)
                tape.record((x,), out, bwd_lazy)
            return out
        r_raw = res_t if res_lazy is None else res_lazy.raw
        r_aff = None if res_lazy is None else (res_lazy.scale, res_lazy.shift)
        # the gated gradient of this output is also the gradient w.r.t. the projection shortcut's BatchNorm output: let the gate reduce against it too
        second = (res_lazy.raw, res_lazy.mean, res_lazy.invstd) if (res_lazy is not None and _FUSE_SHORTCUT_GATE and _FUSE_BN_BWD) else None
        if defer:
            y = hold = LazySum(x, scale, shift, mean, invstd, r_raw, r_aff, want_mask=tape is not None, second=second)
            mask = None
        else:
            y, mask = ops.bn_apply(x, scale, shift, relu=relu, residual=r_raw, res_affine=r_aff, want_mask=tape is not None)
    else:
        y, mean, invstd, mask = ops.bn_train_fwd(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                 relu=relu, residual=res_t, eps=bn.eps, momentum=bn.momentum,
                                                 want_mask=True, skip_mask=tape is None,   # 1 byte per 4 elements for the backward
                                                 partials=partials)
        _bn_order_record(bn, x)
    if tape is not None:
        slot = tape.slot

        if relu and mask is not None:
            y._bn_gate = ops.BnGateCtx(x, mean, invstd, mask=mask, second=second)

        def bwd(dy, existing):
            y_, mask_ = (y, mask) if hold is None else (hold.t, hold.mask)      # a deferred output was written by its first consumer
            part = dy.__dict__.pop("_gate_partials", None)        # consumed here: dy's buffer goes on as the residual gradient
            part_res = dy.__dict__.pop("_gate_partials_res", None)
            if part is not None and part_res is not None and residual is not None and existing[1] is None:
                dy._gate_partials = part_res      # dy IS the gradient w.r.t. the projection shortcut's BatchNorm output: its backward finds its sums
            if part is not None and existing[0] is None and tape.last[0] and getattr(x, "_lazy_dy_ok", False):
                # the second half of the backward is formed by the producing convolution's wgrad / dgrad while they stage it
                dx = ops.LazyGrad(dy, x, ops.bn_bwd_coef(x, bn.weight, mean, invstd, part, grad_of(bn.weight, slot), grad_of(bn.bias, slot)))
                dres = dy if residual is not None else None
            elif part is not None:                # dy arrives relu-gated with its partial sums: no reduction pass, and dresidual IS dy
                dx = ops.bn_bwd_from_partials(dy, x, bn.weight, mean, invstd, part, grad_of(bn.weight, slot), grad_of(bn.bias, slot))
                dres = dy if residual is not None else None
            else:
                dx, dres = ops.bn_train_bwd(dy, y_, x, bn.weight, mean, invstd, relu, grad_of(bn.weight, slot), grad_of(bn.bias, slot),
                                            want_dres=residual is not None, accumulate=True, relu_mask=mask_)
            if residual is None:
                return (_accum(existing[0], dx), None)
            return (_accum(existing[0], dx), _accum(existing[1], dres))
        tape.record((x, residual), y, bwd)
    return y


_FUSE_STEM_POOL = os.environ.get("SSV_NO_STEM_POOL_FUSION", "0") != "1"    # diagnostic switch
_POOLED_STEM_REDUCE = os.environ.get("SSV_NO_POOLED_STEM_REDUCE", "0") != "1"    # diagnostic switch: the stem BatchNorm backward's sums from a walk over the full-resolution map


def bn_relu_maxpool(tape, x, bn):
    """maxpool(relu(bn(x))) of the image stem (networks/resnet.py:147-148).  With the statistics partials of x at hand this is one pass
    each way: the full-resolution activation and its gradient are never written (ssv_bn_relu_maxpool_fwd / _bwd)."""
    partials = x.__dict__.get("_bn_partials")
    # the fused backward walks a map row by row in strides of RT = 256 / min(C / 4, 256) pixels and needs RT <= W (ssv_bn_relu_maxpool_bwd):
    # narrower maps (inputs below 32 px behind the 7x7 / 2 stem) take the three-kernel path, which - like the reference - accepts any size
    narrow = x.shape[2] < 256 // max(1, min(x.shape[3] // 4, 256))
    if partials is None or not _FUSE_STEM_POOL or not _FUSE_BN_APPLY or narrow:
        return maxpool(tape, batchnorm(tape, x, bn, relu=True))
    x.__dict__.pop("_bn_partials")
    _bn_order_wait(bn, x)
    m, c = ops._rows(x)
    mean, invstd, scale, shift = ops.bn_stats_finalize(m, c, partials, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked,
                                                       eps=bn.eps, momentum=bn.momentum)
    _bn_order_record(bn, x)
    keep = tape is not None and _POOLED_STEM_REDUCE
    y, am, *rest = ops.bn_relu_maxpool_fwd(x, scale, shift, keep_xmax=keep)
    if tape is not None:
        slot = tape.slot
        xmax = rest[0] if keep else None            # the conv output at every window's arg-max: the backward's reduction reads 2 x 1/4 of the map instead of all of it

        def bwd(dy, existing):
            dx = ops.bn_relu_maxpool_bwd(dy, am, x, bn.weight, mean, invstd, scale, shift, grad_of(bn.weight, slot), grad_of(bn.bias, slot), xmax=xmax)
            return (_accum(existing[0], dx),)
        tape.record((x,), y, bwd)
    return y


def maxpool(tape, x):
    y, am = ops.maxpool_fwd(x)
    if tape is not None:
        tape.record((x,), y, lambda dy, ex: (_accum(ex[0], ops.maxpool_bwd(dy, am, x.shape)),))
    return y


def global_avgpool(tape, x):
    y = ops.gap_fwd(_tensor(x))
    if tape is not None:
        tape.record((x,), y, lambda dy, ex: (_accum(ex[0], ops.gap_bwd(dy, x.shape)),))
    return y


def linear(tape, x, weight, bias, addend=None):
    """x [M,Din] -> [M,Dout] (+ addend [M,Dout]): a 1x1 'convolution' over a 1x1 image on the same MFMA kernels."""
    b, din = x.shape
    x4 = x.view(b, 1, 1, din)
    a4 = None if addend is None else addend.view(b, 1, 1, -1)
    y4 = ops.conv2d_fwd(x4, weight, 1, 0, bias=bias, addend=a4)
    y = y4.view(b, weight.shape[0])
    if tape is not None:
        need_dx = tape.needs_grad(x)

        slot = tape.slot

        def bwd(dy, existing):
            dy4 = dy.view(b, 1, 1, -1)
            ops.conv2d_wgrad(x4, dy4, weight, grad_of(weight, slot), 1, 0, accumulate=True, dbias=None if bias is None else grad_of(bias, slot))
            dadd = None if addend is None else _accum(existing[1], dy)      # pass-through (dy is not needed again by its producer)
            if not need_dx:
                return (None, dadd)
            ex = existing[0]
            ex4 = None if ex is None else ex.view(b, 1, 1, din)
            dx = ops.conv2d_dgrad(dy4, weight, x4.shape, 1, 0, addend=ex4, out=ex4)
            return (dx.view(b, din), dadd)
        tape.record((x, addend), y, bwd)
    return y


def layernorm(tape, x, ln, addend=None):
    """LayerNorm(x) * weight + bias (+ addend) over the last axis of [M, C]."""
    y, mean, invstd = ops.layernorm_fwd(x, ln.weight, ln.bias, addend, ln.eps)
    if tape is not None:
        slot = tape.slot

        def bwd(dy, existing):
            dx = ops.layernorm_bwd(dy, x, ln.weight, mean, invstd, grad_of(ln.weight, slot), grad_of(ln.bias, slot),
                                   dx_addend=existing[0], accumulate=True)
            return (dx, None if addend is None else _accum(existing[1], dy))
        tape.record((x, addend), y, bwd)
    return y


def gelu(tape, x):
    y = ops.gelu_fwd(x)
    if tape is not None:
        tape.record((x,), y, lambda dy, ex: (_accum(ex[0], ops.gelu_bwd(x, dy)),))
    return y


def ffn_gelu(tape, x, w1, b1, w2, b2, addend=None):
    """fc2(gelu(fc1(x))) (+ addend) with the GELU folded into the GEMM epilogues: fc1 writes the pre-activation and its GELU in
    one pass, and in the backward fc2's dgrad applies gelu' before it stores - no stand-alone element-wise pass in either direction."""
    m, din = x.shape
    inter = w1.shape[0]
    if din % 32 or inter < 128 or inter % 32 or w2.shape[0] % 32 or b1 is None or b2 is None:
        return linear(tape, gelu(tape, linear(tape, x, w1, b1)), w2, b2, addend)
    # with a backward to come the epilogue writes gelu'(h) in the pre-activation's place (nothing downstream reads h itself): the backward multiplies by it;
    # the teacher's pass keeps neither
    dact_form = tape is not None and ops.can_gelu_dact(w1.shape, w2.shape)
    if dact_form:
        h, act = ops.linear_gelu_fwd_dact(x, w1, b1)
    else:
        h, act = ops.linear_gelu_fwd(x, w1, b1, keep_h=tape is not None)
    a4 = None if addend is None else addend.view(m, 1, 1, -1)
    y = ops.conv2d_fwd(act.view(m, 1, 1, inter), w2, 1, 0, bias=b2, addend=a4).view(m, w2.shape[0])
    if tape is not None:
        need_dx = tape.needs_grad(x)
        slot = tape.slot

        def bwd(dy, existing):
            dy4 = dy.view(m, 1, 1, -1)
            ops.conv2d_wgrad(act.view(m, 1, 1, inter), dy4, w2, grad_of(w2, slot), 1, 0, accumulate=True, dbias=grad_of(b2, slot))
            dh = ops.linear_dgrad_mul(dy, w2, h) if dact_form else ops.linear_dgrad_gelu(dy, w2, h)
            x4, dh4 = x.view(m, 1, 1, din), dh.view(m, 1, 1, inter)
            ops.conv2d_wgrad(x4, dh4, w1, grad_of(w1, slot), 1, 0, accumulate=True, dbias=grad_of(b1, slot))
            dadd = None if addend is None else _accum(existing[1], dy)
            if not need_dx:
                return (None, dadd)
            ex = existing[0]
            ex4 = None if ex is None else ex.view(m, 1, 1, din)
            return (ops.conv2d_dgrad(dh4, w1, x4.shape, 1, 0, addend=ex4, out=ex4).view(m, din), dadd)
        tape.record((x, addend), y, bwd)
    return y


def attention(tape, q, k, v, batch, tokens, heads):
    """softmax(q k^T / sqrt(dh)) v per (image, head) on [batch*tokens, heads*dh] matrices; probabilities are never stored."""
    o, lse = ops.attention_fwd(q, k, v, batch, tokens, heads)
    if tape is not None:
        def bwd(dy, existing):
            dq, dk, dv = ops.attention_bwd(q, k, v, o, dy, lse, batch, tokens, heads)
            return (_accum(existing[0], dq), _accum(existing[1], dk), _accum(existing[2], dv))
        tape.record((q, k, v), o, bwd)
    return o


def _stacked(a, b, c):
    """[3*rows, cols] view over three equally shaped matrices when they sit back to back in memory (parameters and their
    gradients do once the optimizer's arena owns them), else None."""
    if a is None or b is None or c is None:
        return None
    nbytes = a.numel() * a.element_size()
    if not (a.is_contiguous() and b.is_contiguous() and c.is_contiguous()):
        return None
    if a.data_ptr() + nbytes != b.data_ptr() or b.data_ptr() + nbytes != c.data_ptr():
        return None
    base = a.untyped_storage()
    if base.data_ptr() != b.untyped_storage().data_ptr() or base.data_ptr() != c.untyped_storage().data_ptr():
        return None                      # adjacent by accident of the allocator, not views of one arena: three GEMMs
    return a.as_strided((3 * a.shape[0], a.shape[1]), (a.shape[1], 1))


def qkv_attention(tape, x, wq, wk, wv, batch, tokens, heads):
    """attention(x Wq^T, x Wk^T, x Wv^T).  With the three weights adjacent in the parameter arena the projections are ONE
    GEMM into an [M, 3*hidden] matrix whose column blocks feed the attention kernel directly (and one wgrad + one dgrad
    in the backward); otherwise three GEMMs - same numbers per element either way."""
    wcat = _stacked(wq.data, wk.data, wv.data)
    if wcat is None:
        q, k, v = linear(tape, x, wq, None), linear(tape, x, wk, None), linear(tape, x, wv, None)
        return attention(tape, q, k, v, batch, tokens, heads)
    m, din = x.shape
    hid = wq.shape[0]
    x4 = x.view(m, 1, 1, din)
    y = ops.conv2d_fwd(x4, wcat, 1, 0).view(m, 3 * hid)
    q, k, v = y[:, :hid], y[:, hid:2 * hid], y[:, 2 * hid:]
    o, lse = ops.attention_fwd(q, k, v, batch, tokens, heads)
    if tape is not None:
        need_dx = tape.needs_grad(x)
        slot = tape.slot

        def bwd(dy, existing):
            dqkv = torch.empty_like(y)
            ops.attention_bwd(q, k, v, o, dy, lse, batch, tokens, heads, out=dqkv)
            d4 = dqkv.view(m, 1, 1, 3 * hid)
            gcat = _stacked(grad_of(wq, slot), grad_of(wk, slot), grad_of(wv, slot))
            if gcat is not None:
                ops.conv2d_wgrad(x4, d4, wcat, gcat, 1, 0, accumulate=True)
            else:
                for i, w in enumerate((wq, wk, wv)):
                    ops.conv2d_wgrad(x4, dqkv[:, i * hid:(i + 1) * hid].contiguous().view(m, 1, 1, hid), w, grad_of(w, slot), 1, 0, accumulate=True)
            if not need_dx:
                return (None,)
            ex = existing[0]
            ex4 = None if ex is None else ex.view(m, 1, 1, din)
            return (ops.conv2d_dgrad(d4, wcat, x4.shape, 1, 0, addend=ex4, out=ex4).view(m, din),)
        tape.record((x,), o, bwd)
    return o


def vit_embed(tape, img_nhwc, cls_weight, pos_weight, patch):
    """[B,H,W,3] -> ([B*T, 3 patch^2 + E] token rows, T); gradients go to the cls and positional embeddings only."""
    tok, t = ops.vit_embed_fwd(img_nhwc, cls_weight, pos_weight, patch)
    if tape is not None:
        slot = tape.slot
        b, p3, e = img_nhwc.shape[0], 3 * patch * patch, pos_weight.shape[1]

        def bwd(dy, existing):
            ops.vit_embed_bwd(dy, b, t, p3, e, grad_of(cls_weight, slot), grad_of(pos_weight, slot), accumulate=True)
            return (None,)
        tape.record((img_nhwc,), tok, bwd)
    return tok, t


def take_cls(tape, x, batch, tokens):
    """Rows 0, T, 2T, ... of the token matrix: the [CLS] embeddings (networks/vit.py:116)."""
    hid = x.shape[1]
    y = x.view(batch, tokens, hid)[:, 0, :].contiguous()
    if tape is not None:
        def bwd(dy, existing):
            dx = existing[0]
            if dx is None:
                dx = ops.fill_(torch.empty_like(x), 0.0)
                dx.view(batch, tokens, hid)[:, 0, :].copy_(dy)
            else:
                dx.view(batch, tokens, hid)[:, 0, :].add_(dy)
            return (dx,)
        tape.record((x,), y, bwd)
    return y


def weightnorm_linear(tape, x, weight_g, weight_v, bias):
    """nn.utils.weight_norm(nn.Linear): w = g * v / ||v||_row, y = x w^T + bias."""
    w, inv = ops.weightnorm_fwd(weight_g, weight_v)
    b, din = x.shape
    x4 = x.view(b, 1, 1, din)
    y = ops.conv2d_fwd(x4, w, 1, 0, bias=bias).view(b, w.shape[0])
    if tape is not None:
        need_dx = tape.needs_grad(x)
        slot = tape.slot

        def bwd(dy, existing):
            dy4 = dy.view(b, 1, 1, -1)
            dw = torch.empty_like(w)
            ops.conv2d_wgrad(x4, dy4, w, dw, 1, 0, accumulate=False)
            ops.weightnorm_bwd(dw, weight_g, weight_v, inv, grad_of(weight_g, slot), grad_of(weight_v, slot), accumulate=True)
            ops.colsum(dy, grad_of(bias, slot), accumulate=True)
            if not need_dx:
                return (None,)
            ex = existing[0]
            ex4 = None if ex is None else ex.view(b, 1, 1, din)
            return (ops.conv2d_dgrad(dy4, w, x4.shape, 1, 0, addend=ex4, out=ex4).view(b, din),)
        tape.record((x,), y, bwd)
    return y


def l2_normalize(tape, x):
    d = x.shape[1]
    zhat, inv = ops.l2norm_fwd(x, normalize=True)
    if tape is not None:
        tape.record((x,), zhat, lambda dy, ex: (_accum(ex[0], ops.l2norm_bwd(zhat, inv, dy.contiguous(), d, True)),))
    return zhat


# ------------------------------------------------------------------------------------------- autograd bridge
class _Bridge(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, x, module, record):
        if not x.is_cuda:
            raise SsvError(f"{type(module).__name__}: the HIP path needs device tensors; there is no CPU fallback")
        if _STREAMS:
            x.record_stream(torch.cuda.current_stream(x.device))      # produced on another stream, read on this one
        xin = module._prepare_input(x.detach())
        tape = Tape(xin, x.requires_grad) if record else None
        sync = getattr(module, "_grad_sync", None)
        if tape is not None and sync is not None:
            tape.sync = sync
            tape.expected = sync.expect(module)  # one more backward pass will report this module's gradient buckets
            stage_mark(tape, module, "rest")     # parameters outside any staged sub-module: complete when the whole tape has run
        y = module._run(tape, xin)
        ctx.tape, ctx.y, ctx.module = tape, y, module            # the tape keys the output by this object (it may be a LazySum)
        yt = _tensor(y)                                          # a module boundary is a consumer like any other: the tensor must exist
        return yt.view_as(yt)      # fresh tensor object for autograd; same storage

    @staticmethod
    def backward(ctx, dy):
        tape = ctx.tape
        if tape is None:
            raise SsvError("backward through a forward that ran under torch.no_grad()")
        ctx.tape = None
        if _STREAMS:
            dy.record_stream(torch.cuda.current_stream(dy.device))
        dx = tape.backward(ctx.y, dy.contiguous())
        if dx is not None:
            dx = ctx.module._finish_input_grad(dx)
        return None, dx, None, None


class HipModule(nn.Module):
    """Base of every module on the HIP path.  Sub-classes implement ``_run(tape, x)``."""

    def __init__(self):
        super().__init__()
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))

    def _prepare_input(self, x):
        return x.contiguous()

    def _finish_input_grad(self, dx):
        return dx

    def grad_stages(self):
        """Data parallel: [[parameters of stage 0], [stage 1], ...] - runs of parameters whose gradients complete together during backward
        (the LAST stage first) - or [] when the module is not staged.  A module that returns stages calls ``stage_mark(tape, self, i)``
        before stage i's forward ops."""
        return []

    def forward(self, x):
        # grad mode is read HERE: inside Function.forward autograd has already switched it off
        return _Bridge.apply(self._anchor, x, self, torch.is_grad_enabled())

    def train(self, mode=True):
        # the reference never leaves train mode; eval-mode BN is not part of the path
        if not mode:
            raise SsvError("eval() is not supported: the reference path always runs BatchNorm with batch statistics")
        return super().train(mode)


# ------------------------------------------------------------------------------------------- parameter holders
# Round 6: on the bf16x3 arithmetic a narrow k x k layer (fewer than 128 output channels: layer1's 3x3) takes the 128 x 64 tile, not the fp32 variants' 256 x 64, and
# is bound by bytes - its input BatchNorm + ReLU formed on load pays there too: +0.6 % images/s, -5.6 GB per step (profiles/r06_probe_narrow_fuse.txt)
_FUSE_NARROW_3X3 = os.environ.get("SSV_NO_NARROW_3X3_INPUT_FUSION", "0") != "1"


class HipConv2d(HipModule):
    """Bias-free convolution; ``weight`` is [O,I,k,k] in channels_last (OHWI) memory."""

    def __init__(self, cin, cout, k, stride=1, pad=0, weight=None, groups=1):
        super().__init__()
        w = torch.empty(cout, cin // groups, k, k) if weight is None else weight
        self.weight = nn.Parameter(w.contiguous(memory_format=torch.channels_last))
        self.stride, self.pad, self.groups = stride, pad, groups

    def has_stats_epilogue(self):
        """True when this convolution's forward leaves the BatchNorm statistics partials of its output (ssv_conv2d_fwd_stats)."""
        return _FUSE_BN_STATS and self.groups == 1 and self.weight.shape[1] % 32 == 0 and self.weight.shape[0] % 4 == 0

    def can_fuse_input(self, x_shape=None):
        """True when this convolution can take a LazyAct (a never-written conv -> BN -> ReLU output) as its input.  ``x_shape`` (NHWC of that input, when the
        caller knows it) lets a narrow k x k layer say yes where it will run Winograd."""
        # a k x k filter re-stages (and re-transforms) every input element k*k times: with the 256 x 64 tile of narrow layers (cout < 128,
        # twice the staged A rows per thread) that VALU work costs more than the apply pass it saves (measured: layer1's 3x3 forward at
        # 99 TFLOP/s fused against 116 on a materialised input, profiles/r02_*_conv_layers_*.csv); wider layers keep the fusion - and so do narrow ones that
        # run Winograd (round 5): the input transform forms BatchNorm + ReLU once per loaded element, whatever the filter size - and, round 6, narrow ones on
        # the bf16x3 arithmetic (_FUSE_NARROW_3X3 above)
        if self.weight.shape[2] > 1 and (not _FUSE_BN_APPLY_3X3 or (self.weight.shape[0] < 128 and not (_FUSE_NARROW_3X3 and ops.ARITHMETIC == "bf16x3"))):
            if not (_FUSE_BN_APPLY_3X3 and _FUSE_NARROW_WINO and x_shape is not None and self.groups == 1
                    and ops.use_winograd(tuple(self.weight.shape), self.stride, self.pad, tuple(x_shape), True)):
                return False
        return _FUSE_BN_APPLY and _FUSE_BN_STATS and ops.can_fuse_conv_input(self.weight.shape[1] * self.groups, self.weight.shape[0], self.groups)

    def _run(self, tape, x, bn_stats=False, compact_dx=False):
        if self.groups > 1:
            return grouped_conv(tape, x, self.weight, self.groups, self.stride, self.pad)
        if self.weight.shape[1] == 3 and (tape is None or not tape.needs_grad(x)) and _PAD_STEM:
            return stem_conv(tape, x, self.weight, self.stride, self.pad, bn_stats=bn_stats)
        return conv(tape, x, self.weight, self.stride, self.pad, bn_stats=bn_stats, compact_dx=compact_dx)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        if not self.weight.data.is_contiguous(memory_format=torch.channels_last):
            self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)
        return self


class HipBatchNorm(HipModule):
    """BatchNorm2d / BatchNorm1d parameters + buffers (train-mode statistics only)."""

    def __init__(self, c, eps=ops.BN_EPS, momentum=ops.BN_MOMENTUM):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.zeros((), dtype=torch.long))
        self.eps, self.momentum = eps, momentum
        object.__setattr__(self, "_order_event", None)

    def _run(self, tape, x):
        return batchnorm(tape, x, self)


class HipLinear(HipModule):
    """``bias=False`` builds a bias-free layer (the ViT's query / key / value projections)."""

    def __init__(self, din, dout, weight=None, bias=None):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(dout, din) if weight is None else weight)
        if bias is False:
            self.register_parameter("bias", None)
        else:
            self.bias = nn.Parameter(torch.empty(dout) if bias is None else bias)

    def _run(self, tape, x, addend=None):
        return linear(tape, x, self.weight, self.bias, addend)


class HipLayerNorm(HipModule):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight, self.bias, self.eps = nn.Parameter(torch.ones(dim)), nn.Parameter(torch.zeros(dim)), eps

    def _run(self, tape, x, addend=None):
        return layernorm(tape, x, self, addend)
