"""GeoMatch, DGCNN variant: drop-in for /root/reference/models/geoMatch_DGCNN.py:12-183 (same constructor,
forward, end_points and parameter names).  Differences from the FFB6D/SplineCNN GeoMatch: both embeddings are
DGCNN edge-conv stacks over `cld_rgb_nrm` / the model buffer, `mesh` is returned as [1,D,M] straight from
the mesh trunk, and the positive radius of the matching loss scales with depth (:66-67)."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .dgcnn import DgcnnMeshEmb, DgcnnPcdEmb
from .geoMatch import pdist
from .layers import PtSeq, pt_conv1d
from .loss import AutomaticWeightedLoss, CircleLoss, FocalLoss


class GeoMatch(nn.Module):
    needs_pyramid = False            # dynamic graphs are built inside the trunks; no neighbour pyramid in the inputs

    def __init__(self, cfg, cls_id, model_points=None):
        super().__init__()
        self.awl = AutomaticWeightedLoss(2)
        self.feat_dim = cfg["feat_dim"]
        self.positive_r = 3
        self.model_emb = DgcnnMeshEmb(cfg, cls_id, model_points=model_points)
        self.pcd_emb = DgcnnPcdEmb(cfg)
        self.circle_loss = CircleLoss(16)
        self.seg_loss_func = FocalLoss(gamma=2)
        self.seg_layer = (PtSeq(self.feat_dim).conv1d(128, bn=True).conv1d(128, bn=True).conv1d(128, bn=True)
                          .conv1d(2, activation=None))
        self.feature_encoding_layer = (PtSeq(self.feat_dim).conv1d(128, bn=True).conv1d(128, bn=True).conv1d(128, bn=True)
                                       .conv1d(self.feat_dim, activation=None, bias=False))
        self.normalize_feature_layer = pt_conv1d(self.feat_dim, self.feat_dim, bn=True)

    def matching_loss(self, similarity, match_idx, mesh_xyz, vis_flag, RT):
        n_node = len(mesh_xyz)
        dev = similarity.device
        idx_in_mesh = match_idx != n_node
        idx_mesh_in = torch.where(idx_in_mesh)[0]
        idx_out_mesh = match_idx == n_node
        vis = vis_flag.to(torch.bool)
        gt_pt = mesh_xyz[match_idx[idx_in_mesh]]
        valid_vis_pts = mesh_xyz[vis]
        dis_matrix = pdist(gt_pt, valid_vis_pts)
        proj = torch.matmul(valid_vis_pts, RT[:, :3].t()) + RT[:, 3:].t()
        positive_radius = self.positive_r / 1000.0 * proj[:, 2]                 # geoMatch_DGCNN.py:66-67
        pts_num, cols = similarity.shape
        p_n_mask = torch.zeros((pts_num, cols - 1), dtype=torch.bool, device=dev)
        p_n_in_mesh = torch.index_select(p_n_mask, 0, idx_mesh_in)
        p_n_in_mesh[:, vis] = dis_matrix < positive_radius
        p_n_mask[idx_in_mesh] = p_n_in_mesh
        p_n_mask = torch.cat([p_n_mask, idx_out_mesh.unsqueeze(1)], dim=1)
        return self.circle_loss(similarity, p_n_mask, 0.2)

    def pointwise_feature_matching(self, rgbd_feature, mesh_feature, x):
        """geoMatch_DGCNN.py:80-135: all points normalised, similarity for the whole batch in one matmul, the
        padding column is the unit vector e0 (not -1 as in the FFB6D variant), rows picked by `origin_labels`."""
        losses = []
        batch = rgbd_feature.shape[0]
        rgbd_feature = F.normalize(rgbd_feature.transpose(1, 2), p=2, dim=2)
        mesh = mesh_feature[0]
        padding = torch.zeros((self.feat_dim, 1), dtype=torch.float32, device=mesh.device)
        padding[0] = 1
        mesh_padded = F.normalize(torch.cat([mesh, padding], dim=1), p=2, dim=0)
        sim = torch.matmul(rgbd_feature, mesh_padded)
        labels, corr, RTs = x["origin_labels"], x["match_idx"], x["RT"]
        mesh_xyz = self.model_emb.mesh[0][:3, :].transpose(0, 1).contiguous()
        for i in range(batch):
            idxs = torch.where(labels[i] == 1)[0]
            if len(idxs) < 3:
                continue
            losses.append(self.matching_loss(sim[i][idxs, :], corr[i].index_select(0, idxs).long(), mesh_xyz,
                                             x["visible_flag"][i], RTs[i]))
        if not losses:
            return torch.zeros((), device=mesh.device)
        return torch.mean(torch.stack(losses))

    def forward(self, inputs, end_points=None):
        if not end_points:
            end_points = {}
        rgbd_emb = self.pcd_emb(inputs["cld_rgb_nrm"])
        mesh_features = self.model_emb()
        rgbd_features = self.feature_encoding_layer(rgbd_emb)
        rgbd_normalized = self.normalize_feature_layer(rgbd_features)
        rgbd_emb = rgbd_emb + rgbd_normalized
        seg_features = self.seg_layer(rgbd_emb)
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
