"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" is RCCL on ROCm; "gloo" for
the CPU rehearsal tests).

Inference / bench: crops are independent, so ranks take disjoint shards and NO data-path collective
runs; the only collective is the max-over-ranks of the wall time.  Training keeps the reference's
semantics (/root/reference/train_lm.py:385-388,412,436-439): SyncBatchNorm + DistributedDataParallel
(gradient all-reduce over RCCL, bucketed and overlapped with backward by DDP).
"""
import os

import torch
import torch.distributed as dist


def env_rank():
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend=None, device=None):
    """env:// rendezvous as the reference (train_lm.py:385-388).  Returns (rank, local_rank, world)."""
    rank, local_rank, world = env_rank()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend=backend, init_method="env://", **kw)
    return rank, local_rank, world


def shard_range(n_items, rank, world):
    """Contiguous, disjoint, covering shards; sizes differ by at most one (first ranks take the remainder)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def max_over_ranks(value, device=None):
    """bench.py timing rule: the slowest rank defines the step time."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def wrap_for_training(model, local_rank=None, sync_bn=True):
    """SyncBN conversion + DDP with find_unused_parameters=True, as train_lm.py:412,436-439.  On one process
    returns the model unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return model
    if sync_bn and next(model.parameters()).is_cuda:
        model = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
    if next(model.parameters()).is_cuda:
        return torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], output_device=local_rank,
                                                         find_unused_parameters=True)
    return torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
