"""Inference-time dense matching of scene descriptors against the object model.

Mirrors the matching part of `cal_frame_poses` (/root/reference/evaluator.py:60-102, lines 78-93):
  seg_res  = argmax(seg_features, dim=0) ; cls_msk = seg_res == 1
  selected = F.normalize(rgbd_features.T[cls_msk]) ; mesh = F.normalize(mesh_features, dim=0)
  obj_pts_sim = selected @ mesh ; max_th, obj_pts_idx = max(obj_pts_sim, dim=1)

The reference runs this per crop on `bs` host threads and materialises the [n_sel, M] matrix.  Here
the whole batch is ONE fused HIP launch sequence (normalise+pack, MFMA similarity with in-register
row arg-max, optional split merge) that never writes the matrix; rows of unselected points are
computed too (the mask is data dependent) and simply ignored by `selected`.
"""
import torch

from . import ops

_PREC = {"bf16x3": ops.MATCH_BF16X3, "f32": ops.MATCH_F32, 0: 0, 1: 1}


def match_frames(end_points, precision="bf16x3", return_sim=False):
    """end_points: GeoMatch.forward output (seg [B,2,N], rgbd [B,128,N], mesh [1,128,M]).
    Returns dict(mask u8[B,N], count i32[B], best_idx i32[B,N], best_sim f32[B,N] [, sim f32[B,N,M]])."""
    seg, rgbd, mesh = end_points["seg"], end_points["rgbd"], end_points["mesh"]
    mask, count = ops.seg_mask(seg)
    out = ops.match(rgbd, mesh[0] if mesh.dim() == 3 else mesh, precision=_PREC[precision], return_sim=return_sim)
    res = dict(mask=mask, count=count, best_idx=out[0], best_sim=out[1])
    if return_sim:
        res["sim"] = out[2]
    return res


def match_tail(end_points, B, N, M, precision=ops.MATCH_BF16X3):
    """The step's tail on packed rows (evaluator.py:78-93): seg mask, descriptor packs, N x M arg-max -> (mask, count, best_idx, best_sim).
    With settings.USE_SIDE_STREAMS the mask -- which the arg-max does not read -- is formed on a side stream beside the matching
    kernel, and the model's descriptor rows are taken from `end_points["mesh_rows"]` when GeoMatch.forward packed them inside its
    mesh fork (same kernels, same operands: same bits)."""
    from . import settings
    seg, rgbd, mesh = end_points["seg"], end_points["rgbd"], end_points["mesh"]
    forked = settings.USE_SIDE_STREAMS and seg.is_cuda and not torch.is_grad_enabled()
    mrows = end_points.get("mesh_rows") if precision == ops.MATCH_BF16X3 else None
    if mrows is None:
        srows, mrows = ops.match_pack2(rgbd, mesh[0] if mesh.dim() == 3 else mesh, precision)      # both packs, one launch
    else:
        srows = ops.match_pack(rgbd, precision)
    if forked:
        with ops.fork(seg.device, 0) as f:                   # side stream 0: behind the segmentation layers, if they are pending there
            f.use(seg)
            mask, count = ops.seg_mask(seg)
        bi, bs = ops.match_packed(srows, mrows, B, N, M, precision)
        f.join(mask, count, seg)
    else:
        mask, count = ops.seg_mask(seg)
        bi, bs = ops.match_packed(srows, mrows, B, N, M, precision)
    return mask, count, bi, bs


def selected(res, b):
    """(obj_pts_idx, max_th) of crop b for the points with seg arg-max == 1, in point order
    (evaluator.py:83-93)."""
    m = res["mask"][b].bool()
    return res["best_idx"][b][m], res["best_sim"][b][m]


def correspondences(res, cld_xyz, model_xyz, b):
    """Scene points and matched model vertices of crop b (evaluator.py:85-99), both [n_sel,3], on the device."""
    m = res["mask"][b].bool()
    return cld_xyz[b][m], model_xyz[res["best_idx"][b][m].long()]
