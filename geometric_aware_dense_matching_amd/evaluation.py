"""Per-instance pose errors for a whole batch on the GPU and the reference's recall tables.

Mirrors /root/reference/evaluator.py:308-463 (`Evaluator._eval_predictions`) with its error functions
  lib/pysixd/pose_error.py:297-337 add / adi, :400-415 re, :425-437 te, :440-445 arp_2d, utils/pose_utils.py:430-454 get_closest_rot
but batched: the reference walks the predictions one instance at a time in numpy; here a batch of estimated and ground-truth poses of
ONE object goes through a handful of tensor operations on the device (ADI's nearest neighbour is the HIP kNN kernel, pose.py), and only the
per-instance scalars come back to the host for the table.  Units as in the reference: metres, degrees, pixels; diameters in metres.
"""
from collections import OrderedDict

import numpy as np
import torch

from . import pose

METRICS = ["ad_2", "ad_5", "ad_10", "ad_0.1", "rete_2", "rete_5", "rete_10", "re_2", "re_5", "re_10", "te_2", "te_5", "te_10",
           "proj_2", "proj_5", "proj_10"]                           # evaluator.py:323-340


def rotation_error_deg(R_est, R_gt):
    """pose_error.py:400-415 for f[n,3,3] batches -> degrees f[n] (computed in fp64: the arccos near 0 needs it)."""
    tr = torch.einsum("nij,nij->n", R_est.double(), R_gt.double())          # trace(R_est R_gt^T)
    cos = (0.5 * (tr.clamp(max=3.0) - 1.0)).clamp(-1.0, 1.0)
    return torch.rad2deg(torch.arccos(cos))


def closest_symmetric_rotation(R_est, R_gt, sym_rots):
    """pose_utils.py:430-454 for batches: among R_gt and R_gt @ S_k the rotation with the smallest error to R_est (the first wins ties,
    R_gt itself before any symmetric copy, as the reference's strict `<` does).  sym_rots f[K,3,3] (model-to-model) or None."""
    if sym_rots is None or len(sym_rots) == 0:
        return R_gt
    S = torch.as_tensor(sym_rots, dtype=R_gt.dtype, device=R_gt.device).reshape(-1, 3, 3)
    cands = torch.cat([R_gt[:, None], torch.einsum("nij,kjl->nkil", R_gt, S)], dim=1)       # [n, 1+K, 3, 3]
    n, k = cands.shape[:2]
    errs = rotation_error_deg(R_est[:, None].expand(n, k, 3, 3).reshape(-1, 3, 3), cands.reshape(-1, 3, 3)).view(n, k)
    best = torch.argmin(errs, dim=1)                                # argmin returns the first minimum
    return cands[torch.arange(n, device=R_gt.device), best]


def reprojection_error_px(RT_est, RT_gt, model_xyz, K):
    """pose_error.py:440-445 (arp_2d): mean pixel distance of the model vertices projected with both poses.  K f[3,3] or f[n,3,3]."""
    Kt = torch.as_tensor(K, dtype=torch.float64, device=RT_est.device)
    Kt = Kt.expand(RT_est.shape[0], 3, 3) if Kt.dim() == 2 else Kt

    def project(RT):
        pc = torch.einsum("nij,nmj->nmi", Kt, pose.transform(model_xyz.double(), RT.double()))
        return pc[..., :2] / pc[..., 2:3]
    return (project(RT_est) - project(RT_gt)).norm(dim=2).mean(dim=1)


def pose_errors(RT_est, RT_gt, model_xyz, K, symmetric=False, sym_rots=None):
    """evaluator.py:378-400 for n instances of one object: RT f32[n,3,4] (model -> camera), model_xyz f32[M,3] (metres) ->
    dict(ad, re, te, proj) of f64[n] on the device.  Symmetric objects: ADI, and re / proj against the closest symmetric ground truth."""
    RT_est = RT_est.float()
    RT_gt = RT_gt.to(RT_est.device).float()
    te = (RT_gt[:, :, 3].double() - RT_est[:, :, 3].double()).norm(dim=1)
    if symmetric:
        R_sym = closest_symmetric_rotation(RT_est[:, :, :3], RT_gt[:, :, :3], sym_rots)
        RT_sym = torch.cat([R_sym, RT_gt[:, :, 3:]], dim=2)
        re = rotation_error_deg(RT_est[:, :, :3], R_sym)
        proj = reprojection_error_px(RT_est, RT_sym, model_xyz, K)
        ad = pose.adi_metric(RT_est, RT_gt, model_xyz).double()
    else:
        re = rotation_error_deg(RT_est[:, :, :3], RT_gt[:, :, :3])
        proj = reprojection_error_px(RT_est, RT_gt, model_xyz, K)
        ad = pose.add_metric(RT_est, RT_gt, model_xyz).double()
    return dict(ad=ad, re=re, te=te, proj=proj)


PRECISION_METRICS = [m for m in METRICS if m != "ad_0.1"]          # evaluator.py:513-529: the precision table has no absolute 10 cm line


class RecallTable:
    """The recall / error bookkeeping of evaluator.py:342-463.  update() takes the errors of a batch of instances of one object,
    missing() records ground truths without a prediction (every recall 0, no error entry: :359-362), table() / format() give the
    reference's table: one line per metric with the per-object mean recall x 100 and the mean over objects, then mean re / te.

    precision=True is `_eval_predictions_precision` (evaluator.py:466-660, "precision as in the DPOD paper"): ground truths without a
    prediction are IGNORED instead of counted as misses (:549-551) and the metric list drops "ad_0.1"; everything else is the same
    bookkeeping.  dump() writes what the reference leaves in its output directory (:449-455 / :647-660)."""

    def __init__(self, precision=False):
        self.precision = bool(precision)
        self.metrics = PRECISION_METRICS if self.precision else METRICS
        self.recalls = OrderedDict()
        self.errors = OrderedDict()

    def _slot(self, obj_name):
        if obj_name not in self.recalls:
            self.recalls[obj_name] = OrderedDict((m, []) for m in self.metrics)
            self.errors[obj_name] = OrderedDict((e, []) for e in ("ad", "re", "te", "proj"))
        return self.recalls[obj_name], self.errors[obj_name]

    def missing(self, obj_name, count=1):
        rec, _ = self._slot(obj_name)
        if self.precision:
            return                                                      # "NOTE: just ignore undetected" (evaluator.py:549-551)
        for m in self.metrics:
            rec[m] += [0.0] * count

    def dump(self, output_dir, dataset_name, method_name=""):
        """errors / recalls as pickles and the table as text, under the reference's file names: `_{dataset}_errors.pkl`,
        `_{dataset}_recalls.pkl`, `_{dataset}_tab.txt` (evaluator.py:449-455); the precision variant prefixes the method name and
        says `precisions` (:647-660).  The reference writes the pickles through mmcv.dump, which is pickle for a .pkl path."""
        import os
        import pickle
        os.makedirs(output_dir, exist_ok=True)
        kind = "precisions" if self.precision else "recalls"
        stem = os.path.join(output_dir, "%s_%s" % (method_name, dataset_name))
        paths = (stem + "_errors.pkl", stem + "_%s.pkl" % kind, stem + ("_tab_precisions.txt" if self.precision else "_tab.txt"))
        with open(paths[0], "wb") as f:
            pickle.dump(self.errors, f)
        with open(paths[1], "wb") as f:
            pickle.dump(self.recalls, f)
        with open(paths[2], "w") as f:
            f.write("%s\n" % self.format())
        return paths


    def update(self, obj_name, errors, diameter):
        """errors: pose_errors() output (tensors or arrays of equal length); diameter in metres."""
        rec, err = self._slot(obj_name)
        e = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)).astype(np.float64) for k, v in errors.items()}
        for k in err:
            err[k] += e[k].tolist()
        ad, re, te, proj = e["ad"], e["re"], e["te"], e["proj"]
        flags = {"ad_2": ad < 0.02 * diameter, "ad_5": ad < 0.05 * diameter, "ad_10": ad < 0.1 * diameter, "ad_0.1": ad < 0.1,
                 "rete_2": (re < 2) & (te < 0.02), "rete_5": (re < 5) & (te < 0.05), "rete_10": (re < 10) & (te < 0.1),
                 "re_2": re < 2, "re_5": re < 5, "re_10": re < 10, "te_2": te < 0.02, "te_5": te < 0.05, "te_10": te < 0.1,
                 "proj_2": proj < 2, "proj_5": proj < 5, "proj_10": proj < 10}           # evaluator.py:408-427
        for m in self.metrics:
            rec[m] += flags[m].astype(np.float64).tolist()

    def table(self):
        obj_names = sorted(self.recalls.keys())
        tab = [["objects"] + obj_names + ["Avg(%d)" % len(obj_names)]]
        for m in self.metrics:
            line, vals = [m], []
            for o in obj_names:
                res = self.recalls[o][m]
                line.append("%.2f" % (100 * np.mean(res)) if len(res) > 0 else 0.0)
                vals.append(np.mean(res) if len(res) > 0 else 0.0)
            if obj_names:
                line.append("%.2f" % (100 * np.mean(vals)))
            tab.append(line)
        for e in ("re", "te"):
            line, vals = [e], []
            for o in obj_names:
                res = self.errors[o][e]
                line.append("%.2f" % np.mean(res) if len(res) > 0 else float("nan"))
                vals.append(np.mean(res) if len(res) > 0 else float("nan"))
            if obj_names:
                line.append("%.2f" % np.mean(vals))
            tab.append(line)
        return tab

    def format(self):
        tab = [[str(c) for c in row] for row in self.table()]
        width = [max(len(r[i]) for r in tab if i < len(r)) for i in range(max(len(r) for r in tab))]
        return "\n".join("  ".join(c.ljust(width[i]) for i, c in enumerate(r)).rstrip() for r in tab)


class BopCsv:
    """The BOP-toolkit result file the reference writes while it walks the predictions (evaluator.py:341,365-373,429-431): header
    `scene_id,im_id,obj_id,score,R,t,time`, one line per predicted instance with R row-major and t in MILLIMETRES, both space
    separated, score and time -1.  `file_name` is the reference's prediction key "scene/…/im_id" (:366-367)."""

    HEADER = "scene_id,im_id,obj_id,score,R,t,time"

    def __init__(self):
        self.lines = [self.HEADER]

    def add(self, file_name, obj_id, R, t, score=-1, time=-1):
        R = np.asarray(R.detach().cpu() if torch.is_tensor(R) else R, dtype=np.float64).reshape(3, 3)
        t = np.asarray(t.detach().cpu() if torch.is_tensor(t) else t, dtype=np.float64).reshape(-1)
        parts = str(file_name).split("/")
        self.lines.append("{scene_id},{im_id},{obj_id},{score},{R},{t},{time}".format(
            scene_id=int(parts[0]), im_id=parts[-1], obj_id=int(obj_id), score=score,
            R=" ".join(map(str, R.flatten().tolist())), t=" ".join(map(str, (t * 1000).flatten().tolist())), time=time))

    def add_batch(self, file_names, obj_id, RT):
        """RT [n,3,4] (pose.solve_poses / infer.run_multi_object output), metres."""
        RT = RT.detach().cpu().double().numpy() if torch.is_tensor(RT) else np.asarray(RT, dtype=np.float64)
        for name, rt in zip(file_names, RT):
            self.add(name, obj_id, rt[:, :3], rt[:, 3])

    def write(self, path):
        import os
        d = os.path.dirname(path)
        if d:
            os.makedirs(d, exist_ok=True)
        with open(path, "w") as f:
            f.write("\n".join(self.lines))                             # no trailing newline, as the reference (:430-431)
        return path
