"""Start one process per GPU from a plain ``python bench.py --gpus N`` / ``SSV_GPUS=N python main.py ...`` invocation.

The reference is single-process (utils/common.py:124-127 is its whole device logic); the data-parallel form of its step
(models/simclr.py:86-95 over a sharded batch) needs N ranks.  The parent found here never becomes a rank: it
starts ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>
<script> <args>`` as a CHILD process, lets the children's stdout / stderr through (rank 0 prints the one JSON line) and
exits with the launcher's return code.

This module is standard library only and must run BEFORE torch is imported: a process that has initialised HIP must
neither fork GPU workers nor replace itself (``os.exec*``) - on the MI355X pool either takes the machine down.  So the
parent stays a thin waiter and ``spawn_ranks`` refuses to run once ``torch`` is in ``sys.modules``.
"""
import os
import signal
import socket
import subprocess
import sys
import time


def in_process_group():
    """True inside a rank started by torch.distributed.run (or any launcher that exports the rendezvous variables)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def rank_command(script, argv, nproc, port=None):
    """The launcher command line for ``nproc`` ranks of ``script argv`` on this node."""
    port = port or int(os.environ.get("MASTER_PORT") or free_port())
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script, *argv]


def spawn_ranks(script, argv, nproc, runner=None):
    """Run ``script argv`` as ``nproc`` ranks in child processes and return the launcher's exit code."""
    if "torch" in sys.modules:
        raise RuntimeError("spawn_ranks must be called before torch is imported: the parent of the ranks must never touch HIP")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // max(1, nproc))))
    env["SSV_LAUNCHED_BY"] = str(os.getpid())
    cmd = rank_command(script, list(argv), nproc)
    return (runner or run_and_forward_signals)(cmd, env=env)


def run_and_forward_signals(cmd, env=None, grace=10.0):
    """``subprocess.call`` for a waiting parent that may itself be told to stop (``timeout -k 10 400 python bench.py --gpus N``, a harness watchdog): the
    child - torch.distributed.run and, below it, the N ranks holding the GPUs and the rendezvous port - runs in its OWN session, and SIGTERM / SIGINT /
    SIGHUP to the parent are passed on to that whole process group (TERM, then KILL after ``grace`` seconds), so no rank outlives the command that
    started it.  Returns the child's exit code (128 + signal when it was stopped this way).  Still no exec, no HIP in this process."""
    child = None
    got = []                                                     # [(signal number, time it arrived)]

    def signal_group(sig):
        if child is None:                                        # a signal between installing the handlers and Popen's return: recorded, passed on below
            return
        try:
            os.killpg(child.pid, sig)                            # the session leader's pid is the group id
        except ProcessLookupError:
            pass

    def forward(signum, _frame):
        got.append((signum, time.monotonic()))
        signal_group(signal.SIGTERM)

    # handlers BEFORE the child exists: a signal in the window between Popen and signal.signal() would otherwise kill this parent by default action and orphan
    # the child's session (the ranks holding the GPUs and the rendezvous port) - the very case this function exists to prevent
    previous = {s: signal.signal(s, forward) for s in (signal.SIGTERM, signal.SIGINT, signal.SIGHUP)}
    try:
        child = subprocess.Popen(cmd, env=env, start_new_session=True)
        if got:
            signal_group(signal.SIGTERM)
        while True:
            try:
                rc = child.wait(timeout=0.25)
                break
            except subprocess.TimeoutExpired:
                if got and time.monotonic() - got[0][1] > grace:
                    signal_group(signal.SIGKILL)
    finally:
        for s, h in previous.items():
            signal.signal(s, h)
    if got:
        signal_group(signal.SIGKILL)                             # whatever is left of the group after its leader has gone
        return 128 + got[0][0]
    return rc


def maybe_spawn_ranks(script, argv, nproc):
    """``nproc`` > 1 outside a process group: become the waiting parent of ``nproc`` ranks and exit with their code.
    Inside a process group (the driver's own torch.distributed.run launch), or for one GPU: return and run in this process."""
    if nproc is None or nproc <= 1 or in_process_group():
        return
    sys.stdout.flush()
    sys.exit(spawn_ranks(script, argv, nproc))


def gpus_flag(argv, default=1):
    """Value of ``--gpus N`` / ``--gpus=N`` in ``argv`` without building the full parser (which lives behind the torch import)."""
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return default
