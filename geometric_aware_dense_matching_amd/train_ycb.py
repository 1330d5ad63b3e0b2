#!/usr/bin/env python3
"""train_ycb.py surface (/root/reference/train_ycb.py: same flags as train_lm.py, dataset_name ycbv, 21 objects
trained as 21 independent jobs by train_ycb.sh:3-9 -- an embarrassingly parallel objects-across-GPUs sharding)."""
from .train_lm import build_parser, main as _main


def main(argv=None):
    import sys
    argv = list(sys.argv[1:] if argv is None else argv)
    if not any(a.startswith("-dataset_name") for a in argv):
        argv.append("-dataset_name=ycbv")
    return _main(argv)


if __name__ == "__main__":
    main()
