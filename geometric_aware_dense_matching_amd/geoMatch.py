"""GeoMatch: drop-in for /root/reference/models/geoMatch.py (`GeoMatch(cfg, cls_id)`,
`forward(inputs, end_points=None) -> dict`), same sub-module and parameter names
(awl, model_emb, pcd_emb, seg_layer, feature_encoding_layer, normalize_feature_layer), same
end_points keys: seg [B,2,N], mesh [1,128,M], rgbd [B,128,N] and, in training mode, loss /
seg_loss / match_loss (geoMatch.py:159-200).

Inputs are the loader's dict (datasets/lm/linemod_pbr.py:572-599) on the GPU.  Index tensors may be
int32 (as the loader makes them) or int64 (as train_lm.py:167-168 widens them).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .ffb6d import FFB6DEmb
from .layers import PtSeq, pt_conv1d
from .loss import AutomaticWeightedLoss, CircleLoss, FocalLoss
from .splinecnn import SplineCNN_Mesh


def pdist(A, B):
    """utils/basic_utils.py:86-89, 'L2' branch."""
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(D2 + 1e-7)


class GeoMatch(nn.Module):
    def __init__(self, cfg, cls_id, model_points=None, cache_mesh_in_eval=False):
        super().__init__()
        self.awl = AutomaticWeightedLoss(2)
        self.feat_dim = cfg["feat_dim"]
        self.positive_r = cfg["neighbor_dis_th"] * cfg["model_d"][cls_id] / 1000.0
        self.model_emb = SplineCNN_Mesh(cfg, cls_id, model_points=model_points)
        self.pcd_emb = FFB6DEmb(cfg["ffb_config"])
        self.circle_loss = CircleLoss(16)
        self.ce_loss = nn.CrossEntropyLoss()
        self.seg_loss_func = FocalLoss(gamma=2)

        self.seg_layer = (PtSeq(self.feat_dim).conv1d(128, bn=True).conv1d(128, bn=True).conv1d(128, bn=True)
                          .conv1d(2, activation=None))
        self.feature_encoding_layer = (PtSeq(128).conv1d(128, bn=True).conv1d(128, bn=True).conv1d(128, bn=True)
                                       .conv1d(self.feat_dim, activation=None, bias=False))
        self.normalize_feature_layer = pt_conv1d(self.feat_dim, self.feat_dim, bn=True)

        self.cache_mesh_in_eval = cache_mesh_in_eval
        self.fused_loss = True          # False: the reference's per-item loop in plain torch (kept for A/B checks)
        self._mesh_cache = None

    # ------------------------------------------------------------------ training matching (geoMatch.py:55-157)
    def matching_loss(self, similarity, match_idx, mesh_xyz, vis_flag, RT):
        n_node = len(mesh_xyz)
        dev = similarity.device
        idx_in_mesh = match_idx != n_node
        idx_mesh_in = torch.where(match_idx != n_node)[0]
        idx_out_mesh = match_idx == n_node
        gt_pt = mesh_xyz[match_idx[idx_in_mesh]]
        vis = vis_flag.to(torch.bool)
        dis_matrix = pdist(gt_pt, mesh_xyz[vis])
        pts_num, cols = similarity.shape
        p_n_mask = torch.zeros((pts_num, cols - 1), dtype=torch.bool, device=dev)
        p_n_in_mesh = torch.index_select(p_n_mask, 0, idx_mesh_in)
        p_n_in_mesh[:, vis] = dis_matrix < self.positive_r
        p_n_mask[idx_in_mesh] = p_n_in_mesh
        p_n_mask = torch.cat([p_n_mask, idx_out_mesh.unsqueeze(1)], dim=1)
        return self.circle_loss(similarity, p_n_mask, 0.2)

    def matching_loss_sys(self, similarity, match_idx, idxs):
        sys_cor = self.model_emb.sys_idx
        pts_num, vert_num = similarity.shape
        cld_idx = torch.arange(pts_num, device=similarity.device)
        cld_idx = torch.cat((cld_idx, cld_idx), dim=0)
        selected_idx = torch.cat((match_idx[idxs], match_idx[sys_cor[idxs]]), dim=0)
        p_n_mask = torch.zeros((pts_num, vert_num), dtype=torch.bool, device=similarity.device)
        p_n_mask[cld_idx, selected_idx] = True
        return self.circle_loss(similarity, p_n_mask, 0.2)

    def pointwise_feature_matching_fused(self, rgbd_feature, mesh_feature, x):
        """Same value and gradients as pointwise_feature_matching (non-symmetric objects), batched:
        one GEMM for all selected points of the batch, then the fused circle-loss rows kernel
        (ops.circle_rows); no per-item Python loop over [n_i, M+1] temporaries."""
        from . import ops
        B = rgbd_feature.shape[0]
        mesh = mesh_feature[0]
        M = mesh.shape[1]
        padding = -torch.ones((self.feat_dim, 1), dtype=torch.float32, device=mesh.device)
        mesh_padded = F.normalize(torch.cat([mesh, padding], dim=1), p=2, dim=0)
        labels = x["labels"]
        sel = labels == 1                                          # [B,N]
        counts = sel.sum(dim=1)
        item_ok = counts >= 3                                      # geoMatch.py:126-127
        sel = sel & item_ok.unsqueeze(1)
        bi, pi = torch.nonzero(sel, as_tuple=True)                 # row-major: item, then point order
        if bi.numel() == 0:
            return torch.zeros((), device=mesh.device)
        rows = F.normalize(rgbd_feature.transpose(1, 2)[bi, pi], p=2, dim=1)      # [R,128]
        sim = torch.matmul(rows, mesh_padded)                      # [R, M+1]
        match = x["match_idx"][bi, pi]
        lrow = ops.circle_rows(sim, match, bi, self.model_emb.xyz.contiguous(), x["visible_flag"], self.positive_r, 16.0, 0.2)
        per_item = torch.zeros(B, dtype=torch.float32, device=mesh.device).index_add_(0, bi, lrow)
        per_item = per_item[item_ok] / counts[item_ok].to(torch.float32)
        return per_item.mean()

    def pointwise_feature_matching(self, rgbd_feature, mesh_feature, x):
        if self.model_emb.sys_corr_idx is None and rgbd_feature.is_cuda and self.fused_loss:
            return self.pointwise_feature_matching_fused(rgbd_feature, mesh_feature, x)
        match_loss = []
        batch = rgbd_feature.shape[0]
        rgbd_feature = rgbd_feature.transpose(1, 2)
        mesh = mesh_feature[0]
        padding = -torch.ones((self.feat_dim, 1), dtype=torch.float32, device=mesh.device)
        mesh_padded = F.normalize(torch.cat([mesh, padding], dim=1), p=2, dim=0)
        labels, corr, RTs = x["labels"], x["match_idx"], x["RT"]
        for i in range(batch):
            idxs = torch.where(labels[i] == 1)[0]
            if len(idxs) < 3:
                continue
            selected_cld = F.normalize(rgbd_feature[i].index_select(0, idxs), p=2, dim=1)
            selected_corr = corr[i].index_select(0, idxs)
            similarity = torch.matmul(selected_cld, mesh_padded)
            if self.model_emb.sys_corr_idx is not None:
                li = self.matching_loss_sys(similarity, corr[i].long(), idxs)
            else:
                li = self.matching_loss(similarity, selected_corr.long(), self.model_emb.xyz.contiguous(),
                                        x["visible_flag"][i], RTs[i])
            match_loss.append(li)
        if len(match_loss) == 0:
            return torch.zeros((), device=mesh.device)
        return torch.mean(torch.stack(match_loss))

    # ------------------------------------------------------------------ forward (geoMatch.py:159-200)
    def mesh_features(self):
        if self.cache_mesh_in_eval and not self.training:
            if self._mesh_cache is None:
                with torch.no_grad():
                    self._mesh_cache = self.model_emb()
            return self._mesh_cache
        self._mesh_cache = None
        return self.model_emb()

    def forward(self, inputs, end_points=None):
        if not end_points:
            end_points = {}
        rgbd_emb = self.pcd_emb(inputs)
        mesh_features = self.mesh_features()
        rgbd_features = self.feature_encoding_layer(rgbd_emb)
        rgbd_normalized = self.normalize_feature_layer(rgbd_features)
        rgbd_emb = rgbd_emb + rgbd_normalized
        seg_features = self.seg_layer(rgbd_emb)
        mesh_features = mesh_features.unsqueeze(0)

        if self.training:
            match_loss = self.pointwise_feature_matching(rgbd_features, mesh_features, inputs)
            seg_loss = self.seg_loss_func(seg_features, inputs["labels"].long())
            end_points["loss"] = self.awl(seg_loss, match_loss)
            end_points["seg_loss"] = seg_loss
            end_points["match_loss"] = match_loss

        end_points["seg"] = seg_features
        end_points["mesh"] = mesh_features
        end_points["rgbd"] = rgbd_features
        return end_points
