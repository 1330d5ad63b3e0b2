"""GPU: batched pose errors and the recall table (geometric_aware_dense_matching_amd/evaluation.py) against the oracle's per-instance
restatement of /root/reference/evaluator.py:308-463 (pinned by the reference-made tests/golden/pose.npz)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.golden import inputs as gin  # noqa: E402

G = os.path.join(os.path.dirname(__file__), "golden")


def _poses(rs, n, spread):
    RT = np.zeros((n, 3, 4), np.float64)
    for i in range(n):
        q, _ = np.linalg.qr(rs.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        RT[i, :, :3] = q
        RT[i, :, 3] = [0.05 * rs.randn(), 0.05 * rs.randn(), 0.8 + 0.1 * rs.rand()]
    est = RT.copy()
    for i in range(n):                                               # estimates from nearly exact to far off
        a = spread[i % len(spread)]
        ax = rs.randn(3)
        ax /= np.linalg.norm(ax)
        Kx = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        dR = np.eye(3) + np.sin(a) * Kx + (1 - np.cos(a)) * Kx @ Kx
        est[i, :, :3] = dR @ RT[i, :, :3]
        est[i, :, 3] += a * 0.05 * rs.randn(3)
    return est, RT


def test_batched_pose_errors_equal_reference_functions_incl_golden():
    from geometric_aware_dense_matching_amd import evaluation
    from geometric_aware_dense_matching_amd.synthetic import LM_K
    from oracle import eval_ref
    g = np.load(os.path.join(G, "pose.npz"))
    pi = gin.pose_inputs()
    sym = gin.sym_rotations()
    model = torch.from_numpy(pi["model"]).cuda()
    est = torch.from_numpy(g["RT"].astype(np.float32)).cuda()
    gt = torch.from_numpy(pi["RT"]).cuda()
    e = evaluation.pose_errors(est, gt, model, LM_K)
    for key, ref, tol in (("ad", g["add"], 1e-5), ("re", g["re"], 2e-3), ("te", g["te"], 1e-5), ("proj", g["proj"], 1e-3)):
        got = e[key].cpu().numpy()                                  # the stored estimate is fp64, the product takes it as fp32
        assert np.allclose(got, ref, rtol=tol, atol=tol), (key, got, ref)
    # symmetric object: ground truth seen through another group element, ADI, closest symmetric rotation
    gt2 = gt.clone()
    for b in range(gt.shape[0]):
        gt2[b, :, :3] = gt[b, :, :3] @ torch.from_numpy(sym[1 + b % (sym.shape[0] - 1)].astype(np.float32)).cuda()
    es = evaluation.pose_errors(est, gt2, model, LM_K, symmetric=True, sym_rots=sym)
    assert np.allclose(es["re"].cpu().numpy(), g["re_sym"], rtol=2e-3, atol=2e-3)
    assert np.allclose(es["proj"].cpu().numpy(), g["proj_sym"], rtol=1e-3, atol=1e-3)
    # a larger random batch against the oracle, both object kinds
    rs = np.random.RandomState(5)
    est_n, gt_n = _poses(rs, 40, spread=(0.002, 0.02, 0.06, 0.3, 2.0))
    pts = pi["model"].astype(np.float64)
    for symmetric in (False, True):
        e = evaluation.pose_errors(torch.from_numpy(est_n.astype(np.float32)).cuda(), torch.from_numpy(gt_n.astype(np.float32)).cuda(), model, LM_K,
                                   symmetric=symmetric, sym_rots=sym if symmetric else None)
        for i in range(est_n.shape[0]):
            a32, g32 = est_n[i].astype(np.float32).astype(np.float64), gt_n[i].astype(np.float32).astype(np.float64)
            ad, re, te, proj = eval_ref.instance_errors(a32[:, :3], a32[:, 3], g32[:, :3], g32[:, 3], pts, LM_K.astype(np.float64),
                                                        sym_info=sym, symmetric=symmetric)
            assert abs(e["ad"][i].item() - ad) < 2e-6 and abs(e["te"][i].item() - te) < 1e-6
            assert abs(e["re"][i].item() - re) < 1e-2 + 1e-4 * re           # fp32 rotation entries: arccos near 0 amplifies their rounding
            assert abs(e["proj"][i].item() - proj) < 1e-3 * max(1.0, proj)


def test_recall_table_equals_reference_bookkeeping():
    from geometric_aware_dense_matching_amd import evaluation
    from geometric_aware_dense_matching_amd.synthetic import LM_K
    from oracle import eval_ref
    pi = gin.pose_inputs()
    model = torch.from_numpy(pi["model"]).cuda()
    pts = pi["model"].astype(np.float64)
    sym = gin.sym_rotations()
    rs = np.random.RandomState(9)
    tab = evaluation.RecallTable()
    rec, err = {}, {}
    spec = {"ape": (0.102, False, 17), "eggbox": (0.165, True, 9), "cat": (0.154, False, 5)}
    for name, (diam, symmetric, n) in spec.items():
        est_n, gt_n = _poses(rs, n, spread=(0.002, 0.01, 0.05, 0.2, 1.0))
        e = evaluation.pose_errors(torch.from_numpy(est_n.astype(np.float32)).cuda(), torch.from_numpy(gt_n.astype(np.float32)).cuda(), model, LM_K,
                                   symmetric=symmetric, sym_rots=sym if symmetric else None)
        tab.update(name, e, diam)
        rec[name] = {m: [] for m in eval_ref.METRICS}
        err[name] = {"re": [], "te": []}
        for i in range(n):
            a32, g32 = est_n[i].astype(np.float32).astype(np.float64), gt_n[i].astype(np.float32).astype(np.float64)
            ad, re, te, proj = eval_ref.instance_errors(a32[:, :3], a32[:, 3], g32[:, :3], g32[:, 3], pts, LM_K.astype(np.float64),
                                                        sym_info=sym, symmetric=symmetric)
            for m, v in eval_ref.recall_flags(ad, re, te, proj, diam).items():
                rec[name][m].append(v)
            err[name]["re"].append(re)
            err[name]["te"].append(te)
    tab.missing("cat", 2)                                            # two ground truths without a prediction (evaluator.py:359-362)
    for m in eval_ref.METRICS:
        rec["cat"][m] += [0.0, 0.0]
    want = eval_ref.table(rec, err)
    got = tab.table()
    assert got[0] == want[0] and len(got) == len(want) == 19
    for a, b in zip(got[1:17], want[1:17]):
        assert a == b, (a, b)                                        # recall lines: identical strings (thresholded flags)
    for a, b in zip(got[17:], want[17:]):
        assert a[0] == b[0] and all(abs(float(x) - float(y)) <= 0.011 for x, y in zip(a[1:], b[1:]))
    text = tab.format()
    assert text.splitlines()[0].split() == ["objects", "ape", "cat", "eggbox", "Avg(3)"] and "rete_5" in text
