"""ORACLE -- test infrastructure only; never imported by the product package.

numpy restatements of the reference's per-instance pose errors and recall tables:
  /root/reference/lib/pysixd/pose_error.py:400-415 re, :425-437 te, :277-294 transform_pts_Rt_2d, :440-445 arp_2d
  /root/reference/utils/pose_utils.py:430-454 get_closest_rot
  /root/reference/evaluator.py:308-463 _eval_predictions (error selection per object, recall thresholds, the table)
re / te / arp_2d / get_closest_rot are pinned by tests/golden/pose.npz (keys re, te, proj, re_sym), which the golden script produced
by executing those functions from the reference's own source text; the table logic follows evaluator.py line by line."""
from collections import OrderedDict

import numpy as np

from . import pose_ref

METRICS = ["ad_2", "ad_5", "ad_10", "ad_0.1", "rete_2", "rete_5", "rete_10", "re_2", "re_5", "re_10", "te_2", "te_5", "te_10",
           "proj_2", "proj_5", "proj_10"]                           # evaluator.py:323-340


def re(R_est, R_gt):
    trace = np.trace(np.dot(R_est, R_gt.T))
    trace = trace if trace <= 3 else 3
    error_cos = min(1.0, max(-1.0, 0.5 * (trace - 1.0)))
    return np.rad2deg(np.arccos(error_cos))


def te(t_est, t_gt):
    return np.linalg.norm(t_gt.flatten() - t_est.flatten())


def transform_pts_Rt_2d(pts, R, t, K):
    pts_c = K.dot(R.dot(pts.T) + t.reshape((3, 1)))
    return np.stack([pts_c[0] / pts_c[2], pts_c[1] / pts_c[2]], axis=1)


def arp_2d(R_est, t_est, R_gt, t_gt, pts, K):
    return np.linalg.norm(transform_pts_Rt_2d(pts, R_est, t_est, K) - transform_pts_Rt_2d(pts, R_gt, t_gt, K), axis=1).mean()


def get_closest_rot(rot_est, rot_gt, sym_info):
    if sym_info is None:
        return rot_gt
    sym_info = np.asarray(sym_info)
    if sym_info.ndim == 2:
        sym_info = sym_info.reshape((1, 3, 3))
    r_err, closest = re(rot_est, rot_gt), rot_gt
    for i in range(sym_info.shape[0]):
        cand = rot_gt.dot(sym_info[i])
        cur = re(rot_est, cand)
        if cur < r_err:
            r_err, closest = cur, cand
    return closest


def instance_errors(R_pred, t_pred, R_gt, t_gt, pts, K, sym_info=None, symmetric=False):
    """evaluator.py:378-400: (ad, re, te, proj) of one instance; symmetric objects use ADI and the closest symmetric GT rotation."""
    t_error = te(t_pred, t_gt)
    if symmetric:
        R_gt_sym = get_closest_rot(R_pred, R_gt, sym_info)
        r_error = re(R_pred, R_gt_sym)
        proj = arp_2d(R_pred, t_pred, R_gt_sym, t_gt, pts, K)
        ad = pose_ref.adi(R_pred, t_pred, R_gt, t_gt, pts)
    else:
        r_error = re(R_pred, R_gt)
        proj = arp_2d(R_pred, t_pred, R_gt, t_gt, pts, K)
        ad = pose_ref.add(R_pred, t_pred, R_gt, t_gt, pts)
    return ad, r_error, t_error, proj


def recall_flags(ad, r_error, t_error, proj, diameter):
    """evaluator.py:408-427 (deg, m, px)."""
    return OrderedDict([
        ("ad_2", float(ad < 0.02 * diameter)), ("ad_5", float(ad < 0.05 * diameter)), ("ad_10", float(ad < 0.1 * diameter)),
        ("ad_0.1", float(ad < 0.1)),
        ("rete_2", float(r_error < 2 and t_error < 0.02)), ("rete_5", float(r_error < 5 and t_error < 0.05)),
        ("rete_10", float(r_error < 10 and t_error < 0.1)),
        ("re_2", float(r_error < 2)), ("re_5", float(r_error < 5)), ("re_10", float(r_error < 10)),
        ("te_2", float(t_error < 0.02)), ("te_5", float(t_error < 0.05)), ("te_10", float(t_error < 0.1)),
        ("proj_2", float(proj < 2)), ("proj_5", float(proj < 5)), ("proj_10", float(proj < 10))])


def table(recalls, errors):
    """evaluator.py:431-463: header + one line per metric (per-object mean x 100, 2 decimals, then the mean over objects) + the mean
    re / te lines.  recalls[obj][metric] / errors[obj]["re"|"te"] are lists."""
    obj_names = sorted(recalls.keys())
    tab = [["objects"] + obj_names + ["Avg(%d)" % len(obj_names)]]
    for m in METRICS:
        line, vals = [m], []
        for o in obj_names:
            res = recalls[o][m]
            if len(res) > 0:
                line.append("%.2f" % (100 * np.mean(res)))
                vals.append(np.mean(res))
            else:
                line.append(0.0)
                vals.append(0.0)
        if obj_names:
            line.append("%.2f" % (100 * np.mean(vals)))
        tab.append(line)
    for e in ("re", "te"):
        line, vals = [e], []
        for o in obj_names:
            res = errors[o][e]
            if len(res) > 0:
                line.append("%.2f" % np.mean(res))
                vals.append(np.mean(res))
            else:
                line.append(float("nan"))
                vals.append(float("nan"))
        if obj_names:
            line.append("%.2f" % np.mean(vals))
        tab.append(line)
    return tab
