"""Batched pose solve on the GPU from the dense correspondences, and the ADD / ADI pose errors.

Mirrors, for a whole batch at once and without leaving the device:
  /root/reference/evaluator.py:60-102 cal_frame_poses  (points with seg arg-max == 1, matched vertex = arg-max column;
      fewer than 5 correspondences -> the sentinel pose [I | (0,0,-1000)])
  /root/reference/utils/pvn3d_eval_utils_kpls.py:43-77 best_fit_transform (Kabsch with the reflection fix)
  /root/reference/lib/pysixd/pose_error.py:297-337 add, adi (ADI's nearest neighbour = the HIP kNN kernel, K=1)
The reference does this on `bs` host threads with numpy (ThreadPoolExecutor, evaluator.py:294-303).
"""
import torch

from . import _lib, ops
from ._lib import check


def kabsch_stats(res, cld_rgb_nrm, model_xyz):
    """res: matching.match_frames output; cld_rgb_nrm f32[B,9,N] (rows 0-2 = xyz); model_xyz f32[M,3] -> f64[B,16]."""
    mask, best_idx = res["mask"], res["best_idx"]
    B, N = mask.shape
    cld = ops._dev(cld_rgb_nrm, torch.float32, "cld_rgb_nrm")
    model_xyz = ops._dev(model_xyz, torch.float32, "model_xyz")
    out = torch.empty((B, 16), dtype=torch.float64, device=mask.device)
    check(_lib.lib().gdm_kabsch_stats_hip(cld.data_ptr(), cld.stride(0), 1, N, model_xyz.data_ptr(), best_idx.data_ptr(),
                                          mask.data_ptr(), B, N, model_xyz.shape[0], out.data_ptr(), ops._stream()),
          "gdm_kabsch_stats_hip")
    return out


def solve_poses(res, cld_rgb_nrm, model_xyz, min_points=5):
    """-> RT f32[B,3,4] mapping model coordinates to the camera frame, valid bool[B].  Two launches (statistics, fit), no
    host synchronisation."""
    st = kabsch_stats(res, cld_rgb_nrm, model_xyz)
    B = st.shape[0]
    RT = torch.empty((B, 3, 4), dtype=torch.float32, device=st.device)
    valid = torch.empty((B,), dtype=torch.uint8, device=st.device)
    check(_lib.lib().gdm_kabsch_solve_hip(st.data_ptr(), B, int(min_points), RT.data_ptr(), valid.data_ptr(), ops._stream()),
          "gdm_kabsch_solve_hip")
    return RT, valid.bool()


def transform(pts, RT):
    """pts f32[M,3], RT f32[B,3,4] -> f32[B,M,3]."""
    return pts[None] @ RT[:, :, :3].transpose(1, 2) + RT[:, None, :, 3]


def add_metric(RT_est, RT_gt, model_xyz):
    """pose_error.py:297-312 for a batch: mean vertex distance, f32[B]."""
    return (transform(model_xyz, RT_est) - transform(model_xyz, RT_gt)).norm(dim=2).mean(dim=1)


def adi_metric(RT_est, RT_gt, model_xyz):
    """pose_error.py:315-337: for every GT-posed vertex the nearest estimated-pose vertex (exact 1-NN, HIP)."""
    pe, pg = transform(model_xyz, RT_est).contiguous(), transform(model_xyz, RT_gt).contiguous()
    _, d2 = ops.knn_batch(pe, pg, 1, return_d2=True)
    return d2[:, :, 0].clamp(min=0).sqrt().mean(dim=1)
