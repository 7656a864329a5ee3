#!/usr/bin/env python3
"""python main.py -c <yaml> -m <arch> -a <algo> -t <task> [-o out] [-l ckpt_dir] - same flags as the reference.

Data parallel over the GPUs of one node: ``SSV_GPUS=N python main.py ...`` (the flag surface stays the reference's, so the rank count is an
environment variable) starts N ranks as child processes - or start them with an outer ``python -m torch.distributed.run``."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    from ssv_amd import launch as _launch                  # standard library only: the parent of the ranks never touches HIP
    _launch.maybe_spawn_ranks(os.path.abspath(__file__), sys.argv[1:], int(os.environ.get("SSV_GPUS", "1") or 1))

from ssv_amd.main import main  # noqa: E402

if __name__ == "__main__":
    main()
