#!/usr/bin/env python3
"""train_lm.py / train_ycb.py entry-point surface of the reference, on the MI355X path.

Same single-dash flags as /root/reference/train_lm.py:35-86 (`-state {train,test,eval} -cls_id -dataset_name
-checkpoint --gpus --local_rank --gpu --deterministic -weight_decay -bn_momentum -bn_decay -decay_step`), same
checkpoint layout, same optimiser / schedulers (Adam 1e-4, CyclicLR 1e-6..1e-3 triangular, BN momentum decay),
SyncBN + DDP over RCCL.  Launch as the reference does (train_lm.sh:8), e.g.
    python -m torch.distributed.run --nproc_per_node=8 -m geometric_aware_dense_matching_amd.train_lm \\
        --gpus=8 -state=train -dataset_name=lmo -cls_id=1 -checkpoint=train_log/lm/checkpoints/

Two things differ by design:
  * the neighbour pyramid is built on the GPU inside `model_fn_dec` (two launches per batch) instead of 22
    KD-tree calls per crop in the DataLoader workers;
  * BOP dataset loaders are out of scope: `-data synthetic` (default) feeds generated crops
    with the loader's item layout, a real loader can be plugged through `--dataset-factory module:function`.
"""
import argparse
import importlib
import os
import time

import numpy as np
import torch
import torch.nn as nn

from . import matching, pose, pyramid, synthetic
from .checkpoint import load_checkpoint, save_checkpoint
from .config import LM_DIAMETERS, make_model_cfg
from .geoMatch import GeoMatch
from .parallel import init_distributed, wrap_for_training

LM_OBJS = {1: "ape", 5: "can", 6: "cat", 8: "driller", 9: "duck", 10: "eggbox", 11: "glue", 12: "holepuncher"}
bnm_clip = 1e-2


def build_parser():
    p = argparse.ArgumentParser(description="Arg parser")
    p.add_argument("-weight_decay", type=float, default=0)
    p.add_argument("-lr", type=float, default=1e-2)
    p.add_argument("-lr_decay", type=float, default=0.5)
    p.add_argument("-decay_step", type=float, default=2e5)
    p.add_argument("-bn_momentum", type=float, default=0.9)
    p.add_argument("-bn_decay", type=float, default=0.5)
    p.add_argument("-checkpoint", type=str, default=None)
    p.add_argument("-state", type=str, default="eval")
    p.add_argument("-dataset_name", type=str, default="lmo")
    p.add_argument("-cls_id", type=int, default=5)
    p.add_argument("--local_rank", type=int, default=int(os.environ.get("LOCAL_RANK", "0")))
    p.add_argument("-n", "--nodes", default=1, type=int)
    p.add_argument("-g", "--gpus", default=8, type=int)
    p.add_argument("-nr", "--nr", default=0, type=int)
    p.add_argument("--gpu", type=str, default=None)
    p.add_argument("--deterministic", action="store_true")
    # additions (not in the reference)
    p.add_argument("-data", type=str, default="synthetic")
    p.add_argument("--dataset-factory", type=str, default=None, help="module:function(cfg, split) -> torch Dataset")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--batch-size", type=int, default=24)
    p.add_argument("--n-points", type=int, default=4096)
    p.add_argument("--n-mesh", type=int, default=4096)
    p.add_argument("--synthetic-items", type=int, default=256)
    p.add_argument("--log-dir", type=str, default="train_log/lm/checkpoints")
    p.add_argument("--save-every", type=int, default=10)
    return p


class BNMomentumScheduler:
    """models/pytorch_utils.py:486-505."""

    def __init__(self, model, bn_lambda, last_epoch=-1):
        self.model, self.lmbd = model, bn_lambda
        self.step(last_epoch + 1)
        self.last_epoch = last_epoch

    def step(self, epoch=None):
        if epoch is None:
            epoch = self.last_epoch + 1
        self.last_epoch = epoch
        m = self.lmbd(epoch)
        for mod in self.model.modules():
            if isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d, nn.SyncBatchNorm)):
                mod.momentum = m


class SyntheticCrops(torch.utils.data.Dataset):
    """Generated items with the loader's keys (datasets/lm/linemod_pbr.py:572-599): model inputs + labels,
    match_idx (index of the corresponding model vertex, M = 'no correspondence'), visible_flag, RT."""

    def __init__(self, n_items, n_points, n_mesh, seed=0):
        self.n_items, self.n_points, self.n_mesh, self.seed = n_items, n_points, n_mesh, seed

    def __len__(self):
        return self.n_items

    def __getitem__(self, i):
        it = synthetic.make_crop(self.seed * 100003 + i, self.n_points)
        rs = np.random.RandomState(self.seed * 7 + i)
        labels = it["labels"].astype(np.int32)
        match = rs.randint(0, self.n_mesh, size=self.n_points).astype(np.int32)
        match[rs.rand(self.n_points) < 0.1] = self.n_mesh
        it.update(labels=labels, match_idx=match, visible_flag=(rs.rand(self.n_mesh) < 0.6).astype(np.uint8),
                  RT=np.eye(4, dtype=np.float32)[:3])
        return it


def to_device(data, device):
    """model_fn_dec's dtype rules (train_lm.py:158-172); int tensors stay int32 (the HIP ops take them as is)."""
    out = {}
    for k, v in data.items():
        if isinstance(v, np.ndarray):
            v = torch.from_numpy(v)
        if not torch.is_tensor(v):
            out[k] = v
            continue
        if v.dtype in (torch.float32, torch.uint8, torch.float64):
            out[k] = v.float().to(device, non_blocking=True)
        elif v.dtype in (torch.int32, torch.int16, torch.int64):
            out[k] = v.to(torch.int32).to(device, non_blocking=True)
        else:
            out[k] = v.to(device)
    return out


def model_fn_dec(model, data, device):
    cu = to_device(data, device)
    if "cld_nei_idx0" not in cu:                              # pyramid on the GPU (two launches per batch)
        cu.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(cu["cld_rgb_nrm"]), cu["dpt_xyz"]))
    return model(cu), cu


class Trainer:
    """train_lm.py:178-296."""

    def __init__(self, model, optimizer, checkpoint_dir, obj_name, lr_scheduler=None, bnm_scheduler=None, device=None,
                 local_rank=0, save_every=10, log_every=100):
        self.model, self.optimizer = model, optimizer
        self.lr_scheduler, self.bnm_scheduler = lr_scheduler, bnm_scheduler
        self.checkpoint_dir, self.obj_name = checkpoint_dir, obj_name
        self.device, self.local_rank, self.save_every, self.log_every = device, local_rank, save_every, log_every
        self.history = []

    def train(self, start_epoch, n_epochs, train_loader, train_sampler=None, max_iters=None):
        it_total = 0
        for epoch in range(start_epoch, n_epochs):
            if train_sampler is not None:
                train_sampler.set_epoch(epoch)
            sums = np.zeros(3)
            t0 = time.time()
            for it, batch in enumerate(train_loader):
                self.model.train()
                out, _ = model_fn_dec(self.model, batch, self.device)
                loss = out["loss"]
                vals = (loss.item(), out["seg_loss"].item(), float(out["match_loss"].detach()))
                sums += vals
                self.history.append(vals)
                if (it + 1) % self.log_every == 0 and self.local_rank == 0:
                    print("avg_loss:{:.4f} seg: {:.4f} match: {:.4f}  time cost:{:.1f} s".format(
                        *(sums / self.log_every), time.time() - t0))
                    sums[:] = 0
                    t0 = time.time()
                loss.backward()
                self.optimizer.step()
                self.optimizer.zero_grad()
                if self.lr_scheduler is not None:
                    self.lr_scheduler.step()
                if self.bnm_scheduler is not None:
                    self.bnm_scheduler.step()
                it_total += 1
                if max_iters is not None and it_total >= max_iters:
                    return it_total
            if (epoch + 1) % self.save_every == 0 and self.local_rank == 0:
                save_checkpoint(self.model, self.optimizer, epoch, self.checkpoint_dir, self.obj_name)
        return it_total


def make_dataset(args, split):
    if args.dataset_factory:
        mod, fn = args.dataset_factory.split(":")
        return getattr(importlib.import_module(mod), fn)(args, split)
    return SyntheticCrops(args.synthetic_items, args.n_points, args.n_mesh, seed=0 if split == "train" else 1)


def _model_points(args, cls_id):
    path = os.path.join("datasets/lm/linemod/kps", "obj_%06d_fps.npy" % cls_id)
    if os.path.exists(path):
        return np.load(path)
    return synthetic.make_model_points(cls_id, args.n_mesh, LM_DIAMETERS.get(cls_id, 100.0))


def train(args):
    torch.backends.cudnn.benchmark = not args.deterministic
    if args.deterministic:
        torch.manual_seed(args.local_rank)
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    rank, local_rank, world = init_distributed("nccl", device)
    train_ds = make_dataset(args, "train")
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds) if world > 1 else None
    loader = torch.utils.data.DataLoader(train_ds, batch_size=args.batch_size, shuffle=sampler is None, drop_last=True,
                                         num_workers=4, sampler=sampler)
    cfg = make_model_cfg(n_mesh_node=args.n_mesh, num_points=args.n_points)
    model = GeoMatch(cfg, args.cls_id, model_points=_model_points(args, args.cls_id)).to(device)
    optimizer = torch.optim.Adam(model.parameters(), lr=0.0001, weight_decay=args.weight_decay)
    it, start_epoch = -1, 0
    obj_name = LM_OBJS.get(args.cls_id, "obj_%02d" % args.cls_id)
    if args.checkpoint is not None:
        ep = load_checkpoint(model, optimizer, os.path.join(args.checkpoint, obj_name, "geomatch"), device=device)
        if ep is not None:
            start_epoch = ep
    model = wrap_for_training(model, local_rank)
    steps = max(1, args.epochs * len(train_ds) // args.batch_size // 6 // max(world, 1))
    lr_scheduler = torch.optim.lr_scheduler.CyclicLR(optimizer, base_lr=1e-6, max_lr=1e-3, cycle_momentum=False,
                                                     step_size_up=steps, step_size_down=steps, mode="triangular")
    bnm = BNMomentumScheduler(model, lambda i: max(args.bn_momentum * args.bn_decay ** int(i * args.batch_size / args.decay_step),
                                                   bnm_clip), last_epoch=it)
    trainer = Trainer(model, optimizer, args.log_dir, obj_name, lr_scheduler, bnm, device, local_rank, args.save_every)
    trainer.train(start_epoch, args.epochs, loader, sampler)


def test(args):
    """train_lm.py:317-373 without the BOP evaluator: per-object models, batched forward (crops grouped by
    object instead of batch-1 per instance, :298-314) and dense matching; returns the correspondences."""
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    cfg = make_model_cfg(n_mesh_node=args.n_mesh, num_points=args.n_points)
    model = GeoMatch(cfg, args.cls_id, model_points=_model_points(args, args.cls_id), cache_mesh_in_eval=True).to(device)
    if args.checkpoint is not None:
        load_checkpoint(model, None, os.path.join(args.checkpoint, LM_OBJS.get(args.cls_id, "obj_%02d" % args.cls_id), "geomatch"),
                        device=device)
    model.eval()
    loader = torch.utils.data.DataLoader(make_dataset(args, "test"), batch_size=args.batch_size, shuffle=False, num_workers=2)
    results = []
    with torch.no_grad():
        for batch in loader:
            t0 = time.perf_counter()
            ep, cu = model_fn_dec(model, batch, device)
            res = matching.match_frames(ep)
            RT, valid = pose.solve_poses(res, cu["cld_rgb_nrm"], model.model_emb.xyz)        # evaluator.py:94-100, batched on the GPU
            torch.cuda.synchronize()
            results.append(dict(time=time.perf_counter() - t0, count=res["count"].cpu(), best_idx=res["best_idx"].cpu(),
                                best_sim=res["best_sim"].cpu(), mask=res["mask"].cpu(), RT=RT.cpu(), valid=valid.cpu()))
    return results


def main(argv=None):
    args = build_parser().parse_args(argv)
    if args.gpu is not None:
        os.environ["CUDA_VISIBLE_DEVICES"] = args.gpu
    if args.state == "train":
        train(args)
    else:
        res = test(args)
        print("processed %d batches, %.1f ms/batch" % (len(res), 1e3 * float(np.mean([r["time"] for r in res]))))


if __name__ == "__main__":
    main()
