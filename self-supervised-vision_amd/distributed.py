"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm), launched with
torch.distributed.run.  Only the exchange steps the path really has:
  * all-gather of the projected embeddings (ONE call for both views) and of their row log-sum-exps (one call, the loss partial rides
    along) so NT-Xent sees global negatives;
  * all-reduce(SUM) of the gradient arena, per bucket (stem, the four stages, the heads), each bucket launched on a side stream as soon
    as its last weight gradient of the backward pass has been enqueued (layer4 first) - SURVEY 8e "Collective 2".
BatchNorm statistics stay local (per rank, per view) - the reference has no SyncBN.
"""
import os

import torch
import torch.distributed as dist

from . import ops


_FORCE = False      # a world of ONE rank still takes the collective code paths (the single-GPU functional test of the RCCL calls)
_EMU = None         # an emulated world (emulate_world): this ONE process runs the kernel sequence of rank r of W, no process group


class _EmulatedWorld:
    """Rank ``rank`` of ``world`` ranks emulated by one process on one GPU (BASELINE config 3's per-rank step measured where only one device
    exists): every code path of the data-parallel step is the real one - the rank's 2B rows of NT-Xent against all 2*B*world gathered
    columns, the per-bucket slab folds launched from the backward pass on the exchange stream - and only the transport is replaced:
      * ``gather(out, mine)`` stands in for the all-gather: it must fill ``out`` [world*n, ...] (this rank's block ``mine`` at slot
        ``rank``).  Default: every slot receives ``mine`` - the job whose ``world`` ranks hold the SAME shard, for which the emulation is
        exact (every peer's embeddings, row log-sum-exps and loss partial equal this rank's), written as ONE device copy of the
        all-gather's own size;
      * ``reduce(t)`` stands in for the SUM all-reduce.  Default: ``t *= world`` (the sum over ``world`` identical ranks), one pass over
        the bucket on the exchange stream like the collective's own local reduce.
    Tests hand in ``gather`` / ``reduce`` closures that supply REAL peers (other shards run one after the other)."""

    def __init__(self, world, rank, gather=None, reduce=None):
        if not (world >= 1 and 0 <= rank < world):
            raise ValueError(f"emulate_world: rank {rank} of {world}")
        self.world, self.rank = int(world), int(rank)
        self.gather = gather or self._replicate
        self.reduce = reduce or self._times_world

    def _replicate(self, out, mine):
        out.view(self.world, *mine.shape).copy_(mine)

    def _times_world(self, t):
        t.mul_(float(self.world))


def emulate_world(world, rank=0, gather=None, reduce=None):
    """Switch the emulated world on (``world`` ranks, this process is ``rank``) or off (``world`` None); returns the previous setting's
    object (hand it back to ``restore_world``).  Refused while a real process group is initialised."""
    global _EMU
    if world is not None and dist.is_available() and dist.is_initialized():
        raise RuntimeError("emulate_world: a process group is initialised - the emulation replaces it, it cannot sit on top of it")
    prev = _EMU
    _EMU = None if world is None else _EmulatedWorld(world, rank, gather, reduce)
    return prev


def restore_world(prev):
    global _EMU
    _EMU = prev


def emulated():
    return _EMU is not None


def _group_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def is_on():
    return _EMU is not None or _group_on()


def world_size():
    if _EMU is not None:
        return _EMU.world
    return dist.get_world_size() if _group_on() else 1


def rank():
    if _EMU is not None:
        return _EMU.rank
    return dist.get_rank() if _group_on() else 0


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK if WORLD_SIZE > 1."""
    global _FORCE
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("SSV_DIST_FORCE", "0") == "1"          # WORLD_SIZE=1 + SSV_DIST_FORCE=1: every collective runs over one rank
    if (ws <= 1 and not force) or (dist.is_available() and dist.is_initialized()) or _EMU is not None:
        return rank(), world_size()
    _FORCE = force and ws <= 1
    if _FORCE:
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if backend is None:
        backend = os.environ.get("SSV_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        # one process per GPU; the modulo only matters for the single-GPU functional test of this path (gloo, 2 ranks on cuda:0)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    return dist.get_rank(), dist.get_world_size()


def all_gather_rows(buf, rows_per_rank):
    """``buf`` is [world*rows_per_rank, ...] with this rank's block already in place; fill the rest."""
    if not is_on():
        return buf
    mine = buf[rank() * rows_per_rank:(rank() + 1) * rows_per_rank].clone()      # input must not alias the output
    with _timed(buf):
        if _EMU is not None:
            _EMU.gather(buf, mine)
        elif dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(buf, mine)                               # one RCCL all-gather straight into place
        else:
            dist.all_gather([buf[r * rows_per_rank:(r + 1) * rows_per_rank] for r in range(world_size())], mine)
    return buf


# ---- timing of the collectives (bench.py --gpus N: comm_ms_per_step) ------------------------------------------------------------
_TIMED = None          # None = off; else a list of (start event, end event) recorded around every collective on the stream it was issued on


def comm_timing(on):
    """Start (True) / stop (False) recording an event pair around every collective.  Returns the previous records' total in ms."""
    global _TIMED
    total = comm_ms()
    _TIMED = [] if on else None
    return total


def comm_ms():
    """Sum of the recorded collectives' durations (device time between the events; call after a synchronize)."""
    if not _TIMED:
        return 0.0
    return float(sum(a.elapsed_time(b) for a, b in _TIMED))


class _timed:
    def __init__(self, t):
        self.on = _TIMED is not None and t.is_cuda

    def __enter__(self):
        if self.on:
            self.a, self.b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.on:
            self.b.record()
            _TIMED.append((self.a, self.b))
        return False


def all_gather_blocks(out, mine):
    """``out`` [world * n, ...] <- every rank's ``mine`` [n, ...] in rank order: ONE collective (RCCL all-gather straight into place)."""
    if not is_on():
        out.copy_(mine)
        return out
    with _timed(out):
        if _EMU is not None:
            _EMU.gather(out, mine)
        elif dist.get_backend() == "nccl":
            dist.all_gather_into_tensor(out, mine)
        else:
            n = mine.shape[0]
            dist.all_gather([out[r * n:(r + 1) * n] for r in range(world_size())], mine)
    return out


def all_reduce_sum(t):
    if is_on():
        with _timed(t):
            if _EMU is not None:
                _EMU.reduce(t)
            else:
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


class BucketedGradSync:
    """Data parallel gradient exchange (collective 2).  Every rank back-propagates the GLOBAL-mean loss through its own samples,
    so the SUM of the per-rank gradients is the large-batch gradient.  The arena is cut into buckets - the stem, the four stages, the
    heads: contiguous parameter runs - and a bucket is all-reduced as soon as every backward pass of the step (the two views run on two HIP
    streams and accumulate into two gradient slabs) has reported its last weight gradient: on a side stream that waits for the reporting
    streams' events, folds the second slab into the first and issues ONE all-reduce for the bucket, while the backward of the earlier
    stages keeps the compute streams busy.  ``finish()`` (called by the optimizer's step) reduces whatever was not reported and makes the
    update wait for the side stream.  Element by element this is the arithmetic of the single-call form (slab fold, then a SUM over ranks)."""

    def __init__(self, optimizer, modules=(), bucketed=True):
        self.arena = optimizer.arena
        self.bucketed = bucketed
        self.buckets = []                   # [name, lo, hi]
        self.index = {}                     # (id(module), stage) -> bucket number
        self.of_module = {}                 # id(outer module) -> bucket numbers its backward reports
        self.pending, self.events, self.launched = {}, {}, set()
        self.stream = None
        off_of = {id(p): (o, (p.numel() + 63) // 64 * 64) for p, o in zip(self.arena.params, self.arena.offsets)}
        taken = set()

        def add(name, params, key):
            spans = sorted(off_of[id(p)] for p in params if id(p) in off_of)
            if not spans:
                return None
            lo, hi = spans[0][0], spans[-1][0] + spans[-1][1]
            if sum(n for _, n in spans) != hi - lo or any(lo < b[2] and b[1] < hi for b in self.buckets):
                return None                 # not one contiguous, disjoint run of the arena: left to finish()
            self.buckets.append([name, lo, hi])
            self.index[key] = len(self.buckets) - 1
            taken.update(id(p) for p in params)
            return len(self.buckets) - 1

        if bucketed:
            for m in modules:
                mine = []
                for sub in m.modules():
                    stages = sub.grad_stages() if hasattr(sub, "grad_stages") else []
                    for i, params in enumerate(stages):
                        b = add(f"{type(sub).__name__}.stage{i}", params, (id(sub), i))
                        if b is not None:
                            mine.append(b)
                rest = [p for p in m.parameters() if id(p) not in taken]
                b = add(f"{type(m).__name__}.rest", rest, (id(m), "rest"))
                if b is not None:
                    mine.append(b)
                self.of_module[id(m)] = mine
                object.__setattr__(m, "_grad_sync", self)

    # ---- called by the tape ------------------------------------------------------------------------------------------------------
    def has(self, module, stage):
        return (id(module), stage) in self.index

    def expect(self, module):
        """One more backward pass (through the outer module ``module``) will report these buckets; returns them - the pass hands the set back
        to ``ready`` so that only a pass that was counted for a bucket can count it down."""
        mine = tuple(self.of_module.get(id(module), ()))
        for b in mine:
            self.pending[b] = self.pending.get(b, 0) + 1
        return frozenset(mine)

    def ready(self, module, stage, expected=None):
        b = self.index[(id(module), stage)]
        if expected is not None and b not in expected:
            return                           # a staged sub-module reached through a module that never expect()ed this bucket: not this pass's to report
        if b in self.launched or self.pending.get(b, 0) <= 0:
            return
        grad = self.arena.grad
        if grad.is_cuda:
            ev = torch.cuda.Event()
            ev.record()                      # on the view stream that has just enqueued this bucket's last weight gradient
            self.events.setdefault(b, []).append(ev)
        self.pending[b] -= 1
        if self.pending[b] == 0:
            self._launch(b)

    # ---- the exchange ------------------------------------------------------------------------------------------------------------
    def _reduce(self, lo, hi):
        a = self.arena
        ops.add_(a.grad[lo:hi], a.grad_alt[lo:hi])          # fold the second view's slab (fixed order: slab 0 + slab 1)
        all_reduce_sum(a.grad[lo:hi])

    def _launch(self, b):
        _, lo, hi = self.buckets[b]
        grad = self.arena.grad
        if grad.is_cuda:
            if self.stream is None:
                self.stream = torch.cuda.Stream(grad.device)
            for ev in self.events.pop(b, ()):
                self.stream.wait_event(ev)
            with torch.cuda.stream(self.stream):
                self._reduce(lo, hi)
        else:
            self._reduce(lo, hi)
        self.launched.add(b)

    def finish(self):
        """Before the optimizer update: reduce every part of the arena no bucket covered or no backward pass reported, then make the
        current stream wait for the exchange.  Afterwards slab 0 holds the global gradient (slab 1 is spent)."""
        a = self.arena
        done = sorted((self.buckets[b][1], self.buckets[b][2]) for b in self.launched)
        if a.grad.is_cuda and self.stream is not None:
            cur = torch.cuda.current_stream(a.grad.device)
            self.stream.wait_stream(cur)                 # the leftovers below were produced on the compute streams
            with torch.cuda.stream(self.stream):
                self._leftovers(done)
            cur.wait_stream(self.stream)
        else:
            self._leftovers(done)
        self.pending.clear()
        self.events.clear()
        self.launched.clear()

    def _leftovers(self, done):
        pos = 0
        for lo, hi in done + [(self.arena.numel, self.arena.numel)]:
            if lo > pos:
                self._reduce(pos, lo)
            pos = max(pos, hi)


def attach_grad_sync(optimizer, modules=(), bucketed=None):
    """Data parallel: hand the optimizer its gradient exchange.  ``modules``: the bridged modules whose parameters the optimizer owns (their
    stages become the buckets); without them - or with SSV_DIST_BUCKETS=0, a diagnostic switch - the whole arena is one call at step()."""
    if is_on():
        if bucketed is None:
            bucketed = os.environ.get("SSV_DIST_BUCKETS", "1") != "0"
        optimizer.grad_sync = BucketedGradSync(optimizer, modules, bucketed=bucketed and len(modules) > 0)
    return optimizer


def detach_grad_sync(optimizer, modules=()):
    """Undo ``attach_grad_sync`` (the emulated-world leg of bench.py hands the trainer back to its single-GPU form)."""
    optimizer.grad_sync = None
    for m in modules:
        if "_grad_sync" in m.__dict__:
            object.__delattr__(m, "_grad_sync")
    return optimizer


def broadcast_parameters(flat):
    if _group_on():
        dist.broadcast(flat, src=0)
    return flat


def broadcast_object(obj, src=0):
    """Rank ``src``'s python object on every rank (run names, small configuration)."""
    if not _group_on():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def barrier():
    if _group_on():
        dist.barrier()
