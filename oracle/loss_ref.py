"""ORACLE -- test infrastructure only; never imported by the product package.

torch-CPU restatement of the reference's TRAINING matching path, written per batch item the way the reference
writes it (materialised [n_i, M+1] similarity, explicit boolean mask, masked log-sum-exp):

  pdist                       /root/reference/utils/basic_utils.py:86-89 ('L2' branch)
  log_sum_exp, circle_loss    /root/reference/models/loss.py:441-459, 470-494 (gamma = 16, m = 0.2)
  matching_loss               /root/reference/models/geoMatch.py:55-83   (3-D radius test -> positive mask)
  matching_loss_sys           /root/reference/models/geoMatch.py:86-100  (symmetric objects: two positive columns per row)
  pointwise_feature_matching  /root/reference/models/geoMatch.py:102-157

  dgcnn_pointwise_feature_matching   /root/reference/models/geoMatch_DGCNN.py:52-135 (padding column e0, per-item per-vertex radius)

Pinned by tests/golden/losses.npz, losses_sym.npz and dgcnn_losses.npz (values and gradients produced by the imported reference,
tests/golden/make_golden.py); the product's fused HIP kernels are compared with these functions and with the goldens.
"""
import torch
import torch.nn.functional as F


def pdist(A, B):
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(D2 + 1e-7)


def log_sum_exp(inputs, keep):
    """Masked LSE over the last dim; `keep` is 1.0 where the entry takes part (loss.py:441-459)."""
    drop = 1.0 - keep
    s, _ = torch.max(inputs + (-1e7 * drop), dim=-1, keepdim=True)
    off = (inputs - s).masked_fill(drop.to(torch.bool), -float("inf"))
    return (s + off.exp().sum(dim=-1, keepdim=True).log()).squeeze(-1)


def circle_loss(sim, mask, m=0.2, gamma=16.0):
    ap = torch.clamp_min(-sim.detach() + 1 + m, min=0.0).masked_fill(~mask, 0)
    an = torch.clamp_min(sim.detach() + m, min=0.0).masked_fill(mask, 0)
    logit_p = -ap * (sim - (1 - m)) * gamma
    logit_n = an * (sim - m) * gamma
    z = log_sum_exp(logit_p, mask.to(torch.float)) + log_sum_exp(logit_n, (~mask).to(torch.float))
    return F.softplus(z).mean()


def positive_mask(match_idx, mesh_xyz, vis_flag, radius):
    """geoMatch.py:57-79: rows = selected points, columns = M vertices + the 'not on the model' column."""
    n_node = len(mesh_xyz)
    on = match_idx != n_node
    vis = vis_flag.to(torch.bool)
    mask = torch.zeros((len(match_idx), n_node), dtype=torch.bool)
    near = pdist(mesh_xyz[match_idx[on]], mesh_xyz[vis]) < radius
    rows = torch.zeros((int(on.sum()), n_node), dtype=torch.bool)
    rows[:, vis] = near
    mask[on] = rows
    return torch.cat([mask, (~on).unsqueeze(1)], dim=1)


def positive_mask_sys(match_idx_all, sys_idx, idxs, n_cols):
    """geoMatch.py:88-97.  The reference indexes the per-vertex symmetry table with the selected POINT indices and the
    per-point match table with the result -- restated as written."""
    rows = torch.arange(len(idxs))
    mask = torch.zeros((len(idxs), n_cols), dtype=torch.bool)
    mask[rows, match_idx_all[idxs]] = True
    mask[rows, match_idx_all[sys_idx[idxs]]] = True
    return mask


def pointwise_feature_matching(rgbd_feature, mesh_feature, labels, match_idx, visible_flag, mesh_xyz, radius, sys_idx=None):
    """rgbd_feature [B,D,N], mesh_feature [1,D,M] -> scalar loss (mean over the items with >= 3 selected points)."""
    B, D, _ = rgbd_feature.shape
    rgbd = rgbd_feature.transpose(1, 2)
    padding = -torch.ones((D, 1), dtype=torch.float32)
    mesh_padded = F.normalize(torch.cat([mesh_feature[0], padding], dim=1), p=2, dim=0)
    losses = []
    for i in range(B):
        idxs = torch.where(labels[i] == 1)[0]
        if len(idxs) < 3:
            continue
        sel = F.normalize(rgbd[i].index_select(0, idxs), p=2, dim=1)
        sim = torch.matmul(sel, mesh_padded)
        if sys_idx is not None:
            mask = positive_mask_sys(match_idx[i].long(), sys_idx, idxs, sim.shape[1])
        else:
            mask = positive_mask(match_idx[i].index_select(0, idxs).long(), mesh_xyz, visible_flag[i], radius)
        losses.append(circle_loss(sim, mask))
    if not losses:
        return torch.zeros(())
    return torch.mean(torch.stack(losses))


def dgcnn_positive_mask(match_idx, mesh_xyz, vis_flag, RT, positive_r):
    """geoMatch_DGCNN.py:52-74: as positive_mask, but the radius of column v is positive_r / 1000 * z of v posed by RT (:65-66)."""
    n_node = len(mesh_xyz)
    on = match_idx != n_node
    vis = vis_flag.to(torch.bool)
    vis_pts = mesh_xyz[vis]
    proj = torch.matmul(vis_pts, RT[:, :3].t()) + RT[:, 3:].t()
    radius = positive_r / 1000.0 * proj[:, 2]
    near = pdist(mesh_xyz[match_idx[on]], vis_pts) < radius
    mask = torch.zeros((len(match_idx), n_node), dtype=torch.bool)
    rows = torch.zeros((int(on.sum()), n_node), dtype=torch.bool)
    rows[:, vis] = near
    mask[on] = rows
    return torch.cat([mask, (~on).unsqueeze(1)], dim=1)


def dgcnn_pointwise_feature_matching(rgbd_feature, mesh_feature, origin_labels, match_idx, visible_flag, RT, mesh_xyz, positive_r=3):
    """geoMatch_DGCNN.py:80-135: every point normalised, one similarity for the batch against the mesh padded with e0, rows picked by
    origin_labels, items with fewer than 3 rows skipped."""
    B, D, _ = rgbd_feature.shape
    rgbd = F.normalize(rgbd_feature.transpose(1, 2), p=2, dim=2)
    padding = torch.zeros((D, 1), dtype=torch.float32)
    padding[0] = 1
    mesh_padded = F.normalize(torch.cat([mesh_feature[0], padding], dim=1), p=2, dim=0)
    sim = torch.matmul(rgbd, mesh_padded)
    losses = []
    for i in range(B):
        idxs = torch.where(origin_labels[i] == 1)[0]
        if len(idxs) < 3:
            continue
        mask = dgcnn_positive_mask(match_idx[i].index_select(0, idxs).long(), mesh_xyz, visible_flag[i], RT[i], positive_r)
        losses.append(circle_loss(sim[i][idxs, :], mask))
    if not losses:
        return torch.zeros(())
    return torch.mean(torch.stack(losses))
