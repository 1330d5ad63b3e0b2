#!/usr/bin/env python3
"""train_ycb.py surface (/root/reference/train_ycb.py).  The reference's file is train_lm.py with three differences, all of which
live in the dataset configuration here (config.DATASET_CONFIGS["ycbv"]): `import config.ycbv_cfg as cfg` (:18) -- 21 object
diameters, neighbor_dis_th 0.06, key points under datasets/ycbv/ycbv/kps, checkpoints under train_log/ycb/checkpoints, train
batch 8 --, `-dataset_name` defaulting to 'ycbv' (:70), and checkpoints loaded with strict=False (:140).  The 21 objects are 21
independent jobs (train_ycb.sh:3-9): an objects-across-GPUs sharding needs no collective at all."""
from . import train_lm


def build_parser():
    p = train_lm.build_parser()
    p.set_defaults(dataset_name="ycbv")
    return p


def main(argv=None):
    import os
    import numpy as np
    args = build_parser().parse_args(argv)
    if args.gpu is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu
    if args.state == "train":
        return train_lm.train(args)
    res = train_lm.test(args)
    print("processed %d batches, %.1f ms/batch" % (len(res), 1e3 * float(np.mean([r["time"] for r in res]))))
    return res


if __name__ == "__main__":
    main()
