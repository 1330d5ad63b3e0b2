"""The `lib/pointops` operator surface of the reference on HIP kernels (/root/reference/lib/pointops/functions/pointops.py).

The reference ships only this Python wrapper -- its 19 CUDA sources are absent and nothing imports it (SURVEY.md section 2,
component 12) -- so the contract is the wrapper's: function names, argument order, shapes and dtypes.  Every function below has
the wrapper's name and signature and runs on libgdm_hip.so:

    furthestsampling(xyz, m)                      :38-57    gdm_furthestsampling_hip
    gathering(features, idx)                      :59-84    gdm_group_gather_hip (K = 1), scatter-add backward
    nearestneighbor(unknown, known)               :87-109   three nearest neighbours: the exact kNN kernel, K = 3 (returns sqrt(d2), idx)
    interpolation(features, idx, weight)          :112-146  gdm_interpolation_forward / backward_hip
    grouping(features, idx)                       :149-178  gdm_group_gather_hip, scatter-add backward
    grouping_int(features, idx)                   :181-200  the same gather on the int32 bit patterns
    ballquery(radius, nsample, xyz, new_xyz)      :203-225  gdm_ballquery_hip
    featuredistribute(max_xyz, xyz)               :228-249  nearest centre of every point: the exact kNN kernel, K = 1
    featuregather(max_feature, distribute_idx)    :252-284  gather + scatter-add backward
    labelstat_ballrange / labelstat_idx / labelstat_and_ballquery   :287-372  gdm_labelstat_*_hip (+ gdm_ballquery_hip)
    knnquery / knnquery_heap / knnquery_naive / knnquery_exclude     :395-533  the exact kNN kernel (ascending d2, ties by index)
    QueryAndGroup, QueryAndGroupForKPConv, GroupAll                  :536-660  modules over the functions above

Index tensors are int32 (the wrapper's `torch.cuda.IntTensor`); inputs must be CUDA tensors (no CPU fallback).
Where the wrapper leaves a point open (the CUDA kernels are not available to read) the PointNet++ / PointWeb convention it was
written for is followed and said so in the docstring.
"""
from typing import Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check


def furthestsampling(xyz: torch.Tensor, m: int) -> torch.Tensor:
    """xyz (b, n, 3) -> idx (b, m) int32; starts at index 0."""
    return ops.furthestsampling(xyz.contiguous(), m)


def gathering(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (b, c, n), idx (b, m) -> (b, c, m)."""
    return ops.gather_nn(features, idx)


def nearestneighbor(unknown: torch.Tensor, known: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """unknown (b, n, 3), known (b, m, 3) -> dist (b, n, 3) = l2 distance to the three nearest `known` points, idx (b, n, 3)."""
    idx, d2 = ops.knn_batch(known.contiguous(), unknown.contiguous(), 3, return_d2=True)
    return torch.sqrt(d2), idx


class _Interpolation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx, weight):
        features = ops._dev(features, torch.float32, "features")
        weight = ops._dev(weight, torch.float32, "weight")
        idx = ops._idx32(idx, "idx")
        b, c, m = features.shape
        n = idx.shape[1]
        assert idx.shape == (b, n, 3) and weight.shape == (b, n, 3)
        out = torch.empty((b, c, n), dtype=torch.float32, device=features.device)
        check(_lib.lib().gdm_interpolation_forward_hip(b, c, m, n, features.data_ptr(), idx.data_ptr(), weight.data_ptr(), out.data_ptr(),
                                                       ops._stream()), "gdm_interpolation_forward_hip")
        ctx.save_for_backward(idx, weight)
        ctx.m = m
        return out

    @staticmethod
    def backward(ctx, grad_out):
        idx, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous().float()
        b, c, n = grad_out.shape
        g = torch.zeros((b, c, ctx.m), dtype=torch.float32, device=grad_out.device)
        check(_lib.lib().gdm_interpolation_backward_hip(b, c, n, ctx.m, grad_out.data_ptr(), idx.data_ptr(), weight.data_ptr(), g.data_ptr(),
                                                        ops._stream()), "gdm_interpolation_backward_hip")
        return g, None, None


def interpolation(features: torch.Tensor, idx: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """features (b, c, m), idx (b, n, 3), weight (b, n, 3) -> (b, c, n) = sum_k weight * features[.., idx_k]."""
    return _Interpolation.apply(features, idx, weight)


def grouping(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (b, c, n), idx (b, m, nsample) -> (b, c, m, nsample)."""
    return ops.group_gather(features, idx)


def grouping_int(features: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """features (b, c, n) integer, idx (b, m, nsample) -> (b, c, m, nsample) with the dtype of `features` (the wrapper returns
    LongTensor for its int64 payload); the gather moves 32-bit patterns, so values must fit int32."""
    if features.dtype not in (torch.int32, torch.int64):
        raise TypeError("grouping_int: features must be int32 or int64, got %s" % features.dtype)
    as_f = features.to(torch.int32).contiguous().view(torch.float32)
    with torch.no_grad():
        out = ops.group_gather(as_f, idx)
    return out.view(torch.int32).to(features.dtype)


def ballquery(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor) -> torch.Tensor:
    """xyz (b, n, 3), new_xyz (b, m, 3) -> idx (b, m, nsample): the first nsample points (ascending index) with d2 < radius^2;
    remaining slots repeat the first hit, all zero when the ball is empty (PointNet++ convention)."""
    return ops.ballquery(radius, nsample, xyz.contiguous(), new_xyz.contiguous())


def featuredistribute(max_xyz: torch.Tensor, xyz: torch.Tensor) -> torch.Tensor:
    """max_xyz (b, n, 3) centres, xyz (b, m, 3) -> distribute_idx (b, m): index of the nearest centre of every point."""
    return ops.knn_batch(max_xyz.contiguous(), xyz.contiguous(), 1)[:, :, 0].contiguous()


def featuregather(max_feature: torch.Tensor, distribute_idx: torch.Tensor) -> torch.Tensor:
    """max_feature (b, c, n), distribute_idx (b, m) -> (b, c, m); backward sums into (b, c, n)."""
    return ops.gather_nn(max_feature, distribute_idx)


def _labelstat(label_stat):
    if not label_stat.is_cuda:
        raise RuntimeError("label_stat must be a CUDA (HIP) tensor: the geoMatch ops have no CPU fallback")
    return label_stat.to(torch.int32).contiguous()


def labelstat_ballrange(radius: float, xyz: torch.Tensor, new_xyz: torch.Tensor, label_stat: torch.Tensor) -> torch.Tensor:
    """xyz (b, n, 3), new_xyz (b, m, 3), label_stat (b, n, nclass) int -> (b, m, nclass): label counts summed over ALL points
    with d2 < radius^2 of each centre."""
    xyz = ops._dev(xyz, torch.float32, "xyz")
    new_xyz = ops._dev(new_xyz, torch.float32, "new_xyz")
    ls = _labelstat(label_stat)
    b, n, nclass = ls.shape
    m = new_xyz.shape[1]
    out = torch.empty((b, m, nclass), dtype=torch.int32, device=xyz.device)
    check(_lib.lib().gdm_labelstat_ballrange_hip(b, n, m, float(radius), nclass, new_xyz.data_ptr(), xyz.data_ptr(), ls.data_ptr(),
                                                 out.data_ptr(), ops._stream()), "gdm_labelstat_ballrange_hip")
    return out


def labelstat_idx(nsample: int, label_stat: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """label_stat (b, n, nclass) int, idx (b, m, nsample) -> (b, m, nclass): label counts summed over the indexed points."""
    ls = _labelstat(label_stat)
    idx = ops._idx32(idx, "idx")
    b, n, nclass = ls.shape
    m = idx.shape[1]
    assert idx.shape[2] == nsample
    out = torch.empty((b, m, nclass), dtype=torch.int32, device=ls.device)
    check(_lib.lib().gdm_labelstat_idx_hip(b, n, m, nsample, nclass, ls.data_ptr(), idx.data_ptr(), out.data_ptr(), ops._stream()),
          "gdm_labelstat_idx_hip")
    return out


def labelstat_and_ballquery(radius: float, nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor, label_stat: torch.Tensor):
    """-> (new_label_stat (b, m, nclass), idx (b, m, nsample)): the ball query's index list and the label counts of ALL points in
    the ball (the wrapper's outputs of `labelstat_and_ballquery_cuda`; PointWeb computes both in one pass over the points)."""
    return labelstat_ballrange(radius, xyz, new_xyz, label_stat), ballquery(radius, nsample, xyz, new_xyz)


def knnquery(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor = None) -> torch.Tensor:
    """xyz (b, n, 3), new_xyz (b, m, 3) -> idx (b, m, nsample), nearest first (exact, fp32, ties by ascending index)."""
    if new_xyz is None:
        new_xyz = xyz
    return ops.knn_batch(xyz.contiguous(), new_xyz.contiguous(), nsample)


knnquery_heap = knnquery            # :466-493: same contract, another CUDA kernel in the reference
knnquery_naive = knnquery           # :395-432: dense distances + sort in torch


def knnquery_exclude(nsample: int, xyz: torch.Tensor, new_xyz: torch.Tensor = None) -> torch.Tensor:
    """:496-533: neighbours 1 .. nsample of the sorted list (the nearest one -- the point itself when new_xyz is xyz -- dropped)."""
    if new_xyz is None:
        new_xyz = xyz
    return ops.knn_batch(xyz.contiguous(), new_xyz.contiguous(), nsample + 1)[:, :, 1:].contiguous()


class QueryAndGroup(nn.Module):
    """:536-585: ball query (radius given) or kNN grouping of xyz differences (+ features)."""

    def __init__(self, radius=None, nsample=32, use_xyz=True, return_idx=False):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz
        self.return_idx = return_idx

    def _group(self, xyz, new_xyz, features, idx):
        if new_xyz is None:
            new_xyz = xyz
        if idx is None:
            idx = ballquery(self.radius, self.nsample, xyz, new_xyz) if self.radius is not None else knnquery_heap(self.nsample, xyz, new_xyz)
        grouped_xyz = grouping(xyz.transpose(1, 2).contiguous(), idx)                    # (b, 3, m, nsample)
        diff = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is not None:
            grouped = grouping(features, idx)
            new_features = torch.cat([diff, grouped], dim=1) if self.use_xyz else grouped
        else:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            new_features = diff
        return new_features, grouped_xyz, idx

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        new_features, grouped_xyz, idx = self._group(xyz, new_xyz, features, idx)
        if self.return_idx:
            return new_features, grouped_xyz, idx.long()
        return new_features, grouped_xyz


class QueryAndGroupForKPConv(QueryAndGroup):
    """:588-633: the same grouping, always returning the neighbour indices."""

    def __init__(self, radius=None, nsample=32, use_xyz=True, return_group_idx=False):
        super().__init__(radius, nsample, use_xyz)
        self.return_group_idx = return_group_idx

    def forward(self, xyz, new_xyz=None, features=None, idx=None):
        return self._group(xyz, new_xyz, features, idx)


class GroupAll(nn.Module):
    """:636-660."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz, new_xyz, features=None):
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is not None:
            grouped = features.unsqueeze(2)
            return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped
        return grouped_xyz
