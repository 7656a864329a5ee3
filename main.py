#!/usr/bin/env python3
"""python main.py -c <yaml> -m <arch> -a <algo> -t <task> [-o out] [-l ckpt_dir] - same flags as the reference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ssv_amd.main import main  # noqa: E402

if __name__ == "__main__":
    main()
