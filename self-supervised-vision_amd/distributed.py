"""One process per GPU over RCCL (torch.distributed backend "nccl" on ROCm), launched with
torch.distributed.run.  Only the exchange steps the path really has:
  * all-gather of the projected embeddings (and their row log-sum-exps) so NT-Xent sees global negatives;
  * one all-reduce(SUM) of the flat gradient arena before the optimizer update.
BatchNorm statistics stay local (per rank, per view) - the reference has no SyncBN.
"""
import os

import torch
import torch.distributed as dist


_FORCE = False      # a world of ONE rank still takes the collective code paths (the single-GPU functional test of the RCCL calls)


def is_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def world_size():
    return dist.get_world_size() if is_on() else 1


def rank():
    return dist.get_rank() if is_on() else 0


def init_from_env(backend=None):
    """Initialise from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT / LOCAL_RANK if WORLD_SIZE > 1."""
    global _FORCE
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    force = os.environ.get("SSV_DIST_FORCE", "0") == "1"          # WORLD_SIZE=1 + SSV_DIST_FORCE=1: every collective runs over one rank
    if (ws <= 1 and not force) or (dist.is_available() and dist.is_initialized()):
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
    if dist.get_backend() == "nccl":
        dist.all_gather_into_tensor(buf, mine)                                   # one RCCL all-gather straight into place
    else:
        dist.all_gather([buf[r * rows_per_rank:(r + 1) * rows_per_rank] for r in range(world_size())], mine)
    return buf


def all_reduce_sum(t):
    if is_on():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def attach_grad_sync(optimizer):
    """Data parallel: every rank back-propagates the GLOBAL-mean loss through its own samples, so the
    SUM of the per-rank gradients is the large-batch gradient.  One contiguous all-reduce."""
    if is_on():
        optimizer.grad_sync = all_reduce_sum
    return optimizer


def broadcast_parameters(flat):
    if is_on():
        dist.broadcast(flat, src=0)
    return flat


def broadcast_object(obj, src=0):
    """Rank ``src``'s python object on every rank (run names, small configuration)."""
    if not is_on():
        return obj
    box = [obj]
    dist.broadcast_object_list(box, src=src)
    return box[0]


def barrier():
    if is_on():
        dist.barrier()
