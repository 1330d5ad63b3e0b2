#!/usr/bin/env python3
"""train_lm.py / train_ycb.py entry-point surface of the reference, on the MI355X path.

Same single-dash flags as /root/reference/train_lm.py:35-86 (`-state {train,test,eval} -cls_id -dataset_name
-checkpoint --gpus --local_rank --gpu --deterministic -weight_decay -bn_momentum -bn_decay -decay_step`), same
checkpoint layout, same optimiser / schedulers (Adam 1e-4, CyclicLR 1e-6..1e-3 triangular, BN momentum decay),
SyncBN + DDP over RCCL.  Launch as the reference does (train_lm.sh:8), e.g.
    python -m torch.distributed.run --nproc_per_node=8 -m geometric_aware_dense_matching_amd.train_lm \\
        --gpus=8 -state=train -dataset_name=lmo -cls_id=1 -checkpoint=train_log/lm/checkpoints/

`-dataset_name {lmo,ycbv}` selects the configuration the reference imports at module top (config/<name>_cfg.py:
diameters, neighbour radius factor, key-point and checkpoint directories, object list, batch sizes, strict / non-strict
checkpoint loading); `--model-variant dgcnn` builds models/geoMatch_DGCNN.py's GeoMatch instead of the FFB6D + SplineCNN one.

Three things differ by design:
  * the neighbour pyramid is built on the GPU inside `model_fn_dec` (two launches per batch) instead of 22
    KD-tree calls per crop in the DataLoader workers;
  * `test()` keeps one model per object of the dataset like the reference (train_lm.py:331-340) but runs the instances of a
    batch GROUPED per object (infer.run_multi_object) instead of one batch-1 forward per instance (:298-314), with the
    input-independent mesh descriptors cached per object;
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

from . import infer, matching, ops, pose, pyramid, settings, synthetic
from .checkpoint import load_checkpoint, save_checkpoint
from .config import LMO_OBJS as LM_OBJS, dataset_config, make_dgcnn_cfg, make_model_cfg
from .geoMatch import GeoMatch
from .parallel import init_distributed, wrap_for_training

bnm_clip = 1e-2
DEFAULT_DATASET = "lmo"                     # train_ycb.py sets "ycbv" (train_ycb.py:70)


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
    p.add_argument("-dataset_name", type=str, default=DEFAULT_DATASET)
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
    p.add_argument("--model-variant", type=str, default="ffb6d", choices=["ffb6d", "dgcnn"],
                   help="ffb6d = models/geoMatch.py (CNN + RandLA + SplineCNN); dgcnn = models/geoMatch_DGCNN.py")
    p.add_argument("--epochs", type=int, default=50)
    p.add_argument("--batch-size", type=int, default=None, help="default: the dataset's TRAIN_BATCH_SIZE / VAL_BATCH_SIZE")
    p.add_argument("--n-points", type=int, default=4096)
    p.add_argument("--n-mesh", type=int, default=4096)
    p.add_argument("--synthetic-items", type=int, default=256)
    p.add_argument("--log-dir", type=str, default=None, help="default: the dataset's checkpoint directory")
    p.add_argument("--save-every", type=int, default=10)
    p.add_argument("--log-every", type=int, default=100)
    p.add_argument("--max-iters", type=int, default=None)
    p.add_argument("--single-object", action="store_true", help="test: only -cls_id instead of every object of the dataset")
    p.add_argument("--eval-output", dest="eval_output", type=str, default=None,
                   help="test: directory for the evaluator's files -- BOP result csv, error / recall / precision pickles, table text")
    p.add_argument("--graph-batch1", action="store_true",
                   help="test: per-object hipGraph replay for single-instance groups (a batch-1 eager step is launch-bound)")
    p.add_argument("--objects-across-gpus", action="store_true",
                   help="train: the dataset's objects are independent jobs (train_ycb.sh:3-9 runs them one after the other): rank r of a "
                        "torch.distributed.run launch trains objects r, r + world, ... on its own GPU as single-process jobs -- no process "
                        "group, no collective")
    p.add_argument("--graph-train", action="store_true",
                   help="train: capture one iteration (forward + losses + backward + Adam) as a hipGraph and replay it (single process)")
    return p


class BNMomentumScheduler:
    """models/pytorch_utils.py:486-505.  As in the reference the setter matches BatchNorm1d/2d/3d only (:478-481): the reference
    converts to SyncBatchNorm BEFORE it builds the scheduler (train_lm.py:412,449-457), so under DDP the SyncBN layers keep their
    constructor momentum (0.1; 0.99 for RandLA's layers) and only single-process runs see the decay."""

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
            if type(mod) in (nn.BatchNorm1d, nn.BatchNorm2d, nn.BatchNorm3d):
                mod.momentum = m


class SyntheticCrops(torch.utils.data.Dataset):
    """Generated items with the loader's keys (datasets/lm/linemod_pbr.py:572-599): model inputs + labels,
    match_idx (index of the corresponding model vertex, M = 'no correspondence'), visible_flag, RT."""

    def __init__(self, n_items, n_points, n_mesh, seed=0, cls_ids=None, with_ids=True):
        self.n_items, self.n_points, self.n_mesh, self.seed = n_items, n_points, n_mesh, seed
        self.cls_ids = list(cls_ids) if cls_ids else None          # test split: item i is an instance of cls_ids[i % len]
        self.with_ids = with_ids                                    # scene_id / im_id as a BOP test split carries them (evaluator.py:366-367)

    def __len__(self):
        return self.n_items

    def __getitem__(self, i):
        it = synthetic.make_crop(self.seed * 100003 + i, self.n_points)
        rs = np.random.RandomState(self.seed * 7 + i)
        labels = it["labels"].astype(np.int32)
        match = rs.randint(0, self.n_mesh, size=self.n_points).astype(np.int32)
        match[rs.rand(self.n_points) < 0.1] = self.n_mesh
        it.update(labels=labels, origin_labels=labels.copy(), match_idx=match, visible_flag=(rs.rand(self.n_mesh) < 0.6).astype(np.uint8),
                  RT=np.eye(4, dtype=np.float32)[:3])
        if self.cls_ids:
            it["cls_id"] = np.int32(self.cls_ids[i % len(self.cls_ids)])
        if self.with_ids:
            it["scene_id"], it["im_id"] = np.int32(1 + self.seed), np.int32(i)
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
    needs_pyramid = getattr(getattr(model, "module", model), "needs_pyramid", True)       # the DGCNN variant builds its own graphs
    if needs_pyramid and "cld_nei_idx0" not in cu:            # pyramid on the GPU (two launches per batch)
        cu.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(cu["cld_rgb_nrm"]), cu["dpt_xyz"]))
    with ops.batched_bn_counters():                       # the 91 BatchNorm step counters as one multi-tensor add
        return model(cu), cu


class Trainer:
    """train_lm.py:178-296."""

    def __init__(self, model, optimizer, checkpoint_dir, obj_name, lr_scheduler=None, bnm_scheduler=None, device=None,
                 local_rank=0, save_every=10, log_every=100, graphed_step=None):
        self.model, self.optimizer = model, optimizer
        self.graphed_step = graphed_step                     # train_graph.GraphedTrainStep: the iteration as one hipGraph launch
        self.lr_scheduler, self.bnm_scheduler = lr_scheduler, bnm_scheduler
        self.checkpoint_dir, self.obj_name = checkpoint_dir, obj_name
        self.device, self.local_rank, self.save_every, self.log_every = device, local_rank, save_every, log_every
        self.history = []

    def train(self, start_epoch, n_epochs, train_loader, train_sampler=None, max_iters=None):
        """train_lm.py:224-296.  The reference reads three `.item()`s per iteration (:271-273), i.e. three device syncs; here the
        running sums stay on the device and are read once per `log_every` iterations (and once at the end for `history`)."""
        it_total = 0
        dev_hist = []
        try:
            for epoch in range(start_epoch, n_epochs):
                if train_sampler is not None:
                    train_sampler.set_epoch(epoch)
                sums = None
                t0 = time.time()
                for it, batch in enumerate(train_loader):
                    self.model.train()
                    if self.graphed_step is not None:
                        out = self.graphed_step.step(batch)      # forward, backward and optimizer.step() in one launch
                    else:
                        out, _ = model_fn_dec(self.model, batch, self.device)
                    loss = out["loss"]
                    vals = torch.stack([loss.detach().float(), out["seg_loss"].detach().float(),
                                        torch.as_tensor(out["match_loss"], device=loss.device).detach().float()])
                    sums = vals if sums is None else sums + vals
                    dev_hist.append(vals)
                    if (it + 1) % self.log_every == 0:
                        self._flush_history(dev_hist)            # the one host read of this log window; frees the per-iteration scalars
                        if self.local_rank == 0:
                            print("avg_loss:{:.4f} seg: {:.4f} match: {:.4f}  time cost:{:.1f} s".format(
                                *(sums / self.log_every).tolist(), time.time() - t0))
                        sums = None
                        t0 = time.time()
                    if self.graphed_step is None:
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
        finally:
            self._flush_history(dev_hist)

    HISTORY_CAP = 100000          # `history` keeps the most recent iterations' (loss, seg, match) triples

    def _flush_history(self, dev_hist):
        if dev_hist:
            self.history += [tuple(v) for v in torch.stack(dev_hist).cpu().tolist()]
            dev_hist.clear()
            if len(self.history) > self.HISTORY_CAP:
                del self.history[:len(self.history) - self.HISTORY_CAP]


def make_dataset(args, split, cls_ids=None):
    if args.dataset_factory:
        mod, fn = args.dataset_factory.split(":")
        return getattr(importlib.import_module(mod), fn)(args, split)
    return SyntheticCrops(args.synthetic_items, args.n_points, args.n_mesh, seed=0 if split == "train" else 1, cls_ids=cls_ids)


def _model_points(args, ds, cls_id):
    path = os.path.join(ds["model_pth"], "obj_%06d_fps.npy" % cls_id)
    if os.path.exists(path):
        return np.load(path)
    return synthetic.make_model_points(cls_id, args.n_mesh, ds["diameters"][cls_id])       # KeyError: not an object of this dataset


def build_model(args, cls_id, cache_mesh_in_eval=False):
    """GeoMatch(cfg.MODEL, cls_id) of the selected dataset and variant (train_lm.py:410, :332)."""
    ds = dataset_config(args.dataset_name)
    if cls_id not in ds["diameters"]:
        raise KeyError("-cls_id=%d is not an object of -dataset_name=%s (ids %s)" % (cls_id, args.dataset_name, sorted(ds["diameters"])))
    pts = _model_points(args, ds, cls_id)
    if args.model_variant == "dgcnn":
        from .geoMatch_DGCNN import GeoMatch as GeoMatchDGCNN
        return GeoMatchDGCNN(make_dgcnn_cfg(n_mesh_node=args.n_mesh, dataset=args.dataset_name), cls_id, model_points=pts)
    cfg = make_model_cfg(n_mesh_node=args.n_mesh, num_points=args.n_points, dataset=args.dataset_name)
    return GeoMatch(cfg, cls_id, model_points=pts, cache_mesh_in_eval=cache_mesh_in_eval)


def obj_name_of(ds, cls_id):
    return ds["objs"].get(cls_id, "obj_%02d" % cls_id)


def object_shard(cls_ids, rank, world):
    """Objects of rank `rank` under --objects-across-gpus: every world-th object of the sorted id list (disjoint, covering)."""
    return sorted(cls_ids)[rank::world]


def train_objects_across_gpus(args):
    """The zero-communication sharding of BASELINE config 3: the 21 YCB-V (8 LM-O) objects are independent training jobs
    (/root/reference/train_ycb.sh:3-9), so each rank of the launch trains its share of them alone on its GPU.  Returns
    {cls_id: Trainer}."""
    import copy
    from .parallel import env_rank
    rank, local_rank, world = env_rank()
    ds = dataset_config(args.dataset_name)
    mine = object_shard(ds["objs"], rank, world)
    out = {}
    for cid in mine:
        a = copy.copy(args)
        a.cls_id, a.local_rank, a.objects_across_gpus, a._single_process = cid, local_rank, False, True
        print("[rank %d/%d] training object %d (%s)" % (rank, world, cid, obj_name_of(ds, cid)), flush=True)
        out[cid] = train(a)
    return out


def train(args):
    if getattr(args, "objects_across_gpus", False):
        return train_objects_across_gpus(args)
    torch.backends.cudnn.benchmark = not args.deterministic
    if args.deterministic:                                    # train_lm.py:377-381
        torch.backends.cudnn.deterministic = True
        torch.manual_seed(args.local_rank)
    ds = dataset_config(args.dataset_name)
    batch_size = args.batch_size or ds["train_batch_size"]
    log_dir = args.log_dir or ds["checkpoints"]
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    if getattr(args, "_single_process", False):
        rank, local_rank, world = 0, args.local_rank, 1       # one object per rank: no process group (--objects-across-gpus)
    else:
        rank, local_rank, world = init_distributed("nccl", device)
    train_ds = make_dataset(args, "train")
    sampler = torch.utils.data.distributed.DistributedSampler(train_ds) if world > 1 else None
    loader = torch.utils.data.DataLoader(train_ds, batch_size=batch_size, shuffle=sampler is None, drop_last=True,
                                         num_workers=4, sampler=sampler)
    model = build_model(args, args.cls_id).to(device)
    # same update rule as the reference's Adam (train_lm.py:414-416); `fused`: one multi-tensor kernel per step instead of ~25 foreach launches
    optimizer = torch.optim.Adam(model.parameters(), lr=0.0001, weight_decay=args.weight_decay, fused=settings.USE_FUSED_ADAM)
    it, start_epoch = -1, 0
    obj_name = obj_name_of(ds, args.cls_id)
    if args.checkpoint is not None:
        ep = load_checkpoint(model, optimizer, os.path.join(args.checkpoint, obj_name, "geomatch"), device=device, strict=ds["load_strict"])
        if ep is not None:
            start_epoch = ep
    model = wrap_for_training(model, local_rank)
    graphed = None
    if args.graph_train:
        if world > 1:
            raise SystemExit("--graph-train captures a single-process iteration; run multi-rank training without it")
        from .train_graph import GraphedTrainStep
        graphed = GraphedTrainStep(model, optimizer, device)      # before the schedulers: they then drive the device-side lr
    steps = max(1, args.epochs * len(train_ds) // batch_size // 6 // max(world, 1))
    lr_scheduler = torch.optim.lr_scheduler.CyclicLR(optimizer, base_lr=1e-6, max_lr=1e-3, cycle_momentum=False,
                                                     step_size_up=steps, step_size_down=steps, mode="triangular")
    bnm = BNMomentumScheduler(model, lambda i: max(args.bn_momentum * args.bn_decay ** int(i * batch_size / args.decay_step),
                                                   bnm_clip), last_epoch=it)
    trainer = Trainer(model, optimizer, log_dir, obj_name, lr_scheduler, bnm, device,
                      0 if getattr(args, "_single_process", False) else local_rank,     # an objects-across-gpus rank logs / saves its own objects
                      args.save_every, args.log_every, graphed_step=graphed)
    trainer.train(start_epoch, args.epochs, loader, sampler, max_iters=args.max_iters)
    return trainer


def test(args):
    """train_lm.py:317-373 without the BOP evaluator: ONE MODEL PER OBJECT of the dataset (:331-340), every batch dispatched per
    instance by its `cls_id` (:298-314) -- grouped per object into true batches here -- then dense matching and the batched GPU
    pose solve (evaluator.py:78-100).  Returns per-batch results in the loader's instance order."""
    ds = dataset_config(args.dataset_name)
    batch_size = args.batch_size or ds["val_batch_size"]
    device = torch.device("cuda", args.local_rank)
    torch.cuda.set_device(device)
    ids = [args.cls_id] if args.single_object else sorted(ds["objs"])
    ckpt_root = args.checkpoint or ds["checkpoints"]
    model_dict = {}
    for cid in ids:
        model = build_model(args, cid, cache_mesh_in_eval=True).to(device)
        stem = os.path.join(ckpt_root, obj_name_of(ds, cid), "geomatch")
        if os.path.exists(stem + ".pth.tar"):                 # the reference also skips objects without a checkpoint (:336)
            load_checkpoint(model, None, stem, device=device, strict=ds["load_strict"])
        model_dict[cid] = model.eval()
    loader = torch.utils.data.DataLoader(make_dataset(args, "test", cls_ids=ids), batch_size=batch_size, shuffle=False, num_workers=2)
    graphs = {}
    results = []
    # evaluator.py:308-463: when the loader carries ground-truth poses (`RT`), every instance's ADD(-S) / re / te / re-projection error
    # is computed on the device per object group and the reference's recall table is printed at the end
    from . import evaluation
    table = evaluation.RecallTable()
    prec_table = evaluation.RecallTable(precision=True)      # evaluator.py:466-660, written beside the recall table with --eval_output
    bop = evaluation.BopCsv()                                # evaluator.py:341,365-373: one line per predicted instance
    n_seen = 0
    bop_ids_seen = True                                      # every batch so far carried scene_id / im_id
    sym_names = set(ds.get("sym_objs", ()))                 # config/*_cfg.py SYM_OBJS: object NAMES
    with torch.no_grad():
        for batch in loader:
            t0 = time.perf_counter()
            cu = to_device(batch, device)
            cls = cu["cls_id"].cpu().tolist() if "cls_id" in cu else [args.cls_id] * cu["cld_rgb_nrm"].shape[0]
            if args.graph_batch1 and args.model_variant == "ffb6d" and len(cls) == 1:
                cid = cls[0]
                one = {k: cu[k] for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
                if cid not in graphs:
                    graphs[cid] = infer.GraphedPipeline(model_dict[cid], one)
                out = {k: v.clone() for k, v in graphs[cid](one).items()}
            else:
                out = infer.run_multi_object(model_dict, cu, cls)
            torch.cuda.synchronize()
            results.append(dict(time=time.perf_counter() - t0, cls_id=cls, count=out["mask"].sum(dim=1).cpu(), best_idx=out["best_idx"].cpu(),
                                best_sim=out["best_sim"].cpu(), mask=out["mask"].cpu(), RT=out["RT"].cpu(), valid=out["valid"].cpu()))
            if "RT" in cu and cu["RT"].dim() == 3:
                Kcam = cu["K"] if "K" in cu else torch.from_numpy(synthetic.LM_K).to(device)
                for cid in sorted(set(cls)):
                    rows = torch.tensor([i for i, c in enumerate(cls) if c == cid], device=device)
                    err = evaluation.pose_errors(out["RT"][rows], cu["RT"][rows][:, :3], model_dict[cid].model_emb.xyz,
                                                 Kcam[rows] if Kcam.dim() == 3 else Kcam, symmetric=obj_name_of(ds, cid) in sym_names,
                                                 sym_rots=getattr(model_dict[cid].model_emb, "sym_rots", None))
                    table.update(obj_name_of(ds, cid), err, ds["diameters"][cid] / 1000.0)
                    prec_table.update(obj_name_of(ds, cid), err, ds["diameters"][cid] / 1000.0)
            # The reference walks the GROUND-TRUTH annotations (evaluator.py:340-373): a csv line is appended only for a prediction that
            # has a ground-truth entry, keyed by the real "scene/.../im_id" (:366-367), and a ground truth without a prediction counts as a
            # miss in the recall table (:355-363).  So: lines only where the loader supplied RT AND scene_id / im_id (ids invented here
            # would mean nothing to bop_toolkit and change with the batch order -- without them the csv is not written, see below), and
            # `n_undetected` [bs] (ground truths of the instance's object in its image that the detector missed), when the loader
            # carries it, goes to RecallTable.missing.
            has_gt = "RT" in cu and cu["RT"].dim() == 3
            has_ids = "scene_id" in batch and "im_id" in batch
            bop_ids_seen = bop_ids_seen and has_ids
            if has_gt and has_ids:
                for i, cid in enumerate(cls):
                    bop.add("%06d/%06d" % (int(batch["scene_id"][i]), int(batch["im_id"][i])), cid, out["RT"][i, :, :3], out["RT"][i, :, 3])
            if has_gt and "n_undetected" in batch:
                for i, cid in enumerate(cls):
                    miss = int(batch["n_undetected"][i])
                    if miss > 0:
                        table.missing(obj_name_of(ds, cid), miss)
                        prec_table.missing(obj_name_of(ds, cid), miss)      # (the precision variant ignores them: evaluator.py:549-551)
            n_seen += len(cls)
    if table.recalls:
        test.last_table = table
        if args.local_rank == 0:
            print(table.format())
    if getattr(args, "eval_output", None) and args.local_rank == 0:
        # what the reference's evaluator leaves behind: the BOP result csv (:429-431) and, when ground truth was there, the error /
        # recall pickles and the table text of both variants (:449-455, :647-660)
        written = []
        if bop_ids_seen and len(bop.lines) > 1:
            written.append(bop.write(os.path.join(args.eval_output, "%s_%s-test.csv" % (args.model_variant, args.dataset_name))))
        else:
            import warnings
            warnings.warn("train_lm test: the BOP result csv was NOT written: the dataset does not carry `scene_id` / `im_id` (or no "
                          "instance had a ground-truth pose); the reference keys every csv line by them (evaluator.py:366-373)")
        if table.recalls:
            written += list(table.dump(args.eval_output, args.dataset_name + "_test"))
            written += list(prec_table.dump(args.eval_output, args.dataset_name + "_test", method_name=args.model_variant))
        test.last_outputs = written
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
