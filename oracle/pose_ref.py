"""ORACLE -- test infrastructure only; never imported by the product package.

numpy restatements of /root/reference/utils/pvn3d_eval_utils_kpls.py:43-77 (best_fit_transform) and
/root/reference/lib/pysixd/pose_error.py:297-337 (add, adi; misc.transform_pts_Rt :895-905).
Pinned by tests/golden/pose.npz, which the golden script produced by executing those functions from the
reference's own source text."""
import numpy as np
from scipy import spatial


def best_fit_transform(A, B):
    m = A.shape[1]
    cA, cB = np.mean(A, axis=0), np.mean(B, axis=0)
    H = np.dot((A - cA).T, B - cB)
    U, S, Vt = np.linalg.svd(H)
    R = np.dot(Vt.T, U.T)
    if np.linalg.det(R) < 0:
        Vt[m - 1, :] *= -1
        R = np.dot(Vt.T, U.T)
    T = np.zeros((3, 4))
    T[:, :3] = R
    T[:, 3] = cB.T - np.dot(R, cA.T)
    return T


def transform_pts_Rt(pts, R, t):
    return (R.dot(pts.T) + t.reshape((3, 1))).T


def add(R_est, t_est, R_gt, t_gt, pts):
    return np.linalg.norm(transform_pts_Rt(pts, R_est, t_est) - transform_pts_Rt(pts, R_gt, t_gt), axis=1).mean()


def adi(R_est, t_est, R_gt, t_gt, pts):
    pe, pg = transform_pts_Rt(pts, R_est, t_est), transform_pts_Rt(pts, R_gt, t_gt)
    d, _ = spatial.cKDTree(pe).query(pg, k=1)
    return d.mean()
