"""DGCNN embeddings for the geoMatch_DGCNN variant (BASELINE config 4).

Mirrors /root/reference/models/dgcnn.py: `DgcnnPcdEmb` :58-136 (cloud, k=16) and `DgcnnMeshEmb` :138-237
(object model, k=20).  Parameter names are identical (`bn1..bn8` AND their aliases `conv1.1 ...` inside the
Sequentials, `conv9`, buffer `mesh`).  The dynamic graph (dgcnn.py:21-56) is rebuilt three times per
forward: dense negative squared distances by one GEMM (same formula as the reference), row-wise top-k by a
HIP kernel, edge features cat(x_j - x_i, x_i) by a HIP kernel; the reference's hard-coded
`torch.device('cuda')` (:39) is gone.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from . import ops
from .layers import folded_bn, fused_eval
from .synthetic import COLOR_MEAN, COLOR_STD_MESH


def knn(x, k):
    """dgcnn.py:21-27: x f32[B,C,n] -> idx i32[B,n,k], nearest first (largest negative squared distance)."""
    gram = torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1)
    # pairwise_distance = -xx - (-2 * gram) - xx^T is formed inside the top-k kernel (same operations, same order)
    return ops.topk_negdist(gram, xx, k)


def get_graph_feature(x, k=20, idx=None, dim9=False):
    """dgcnn.py:30-56 -> f32[B,2C,n,k]."""
    x = x.contiguous()
    if idx is None:
        idx = knn(x[:, :3].contiguous(), k) if dim9 else knn(x, k)
    return ops.edge_feature(x, idx)


def _lrelu():
    return nn.LeakyReLU(negative_slope=0.2)


class _DgcnnTrunk(nn.Module):
    def _build(self, embed_dim, feat_dim, dropout):
        self.bn1 = nn.BatchNorm2d(64)
        self.bn2 = nn.BatchNorm2d(64)
        self.bn3 = nn.BatchNorm2d(64)
        self.bn4 = nn.BatchNorm2d(64)
        self.bn5 = nn.BatchNorm2d(64)
        self.bn6 = nn.BatchNorm1d(embed_dim)
        self.bn7 = nn.BatchNorm1d(512)
        self.bn8 = nn.BatchNorm1d(256)
        self.conv1 = nn.Sequential(nn.Conv2d(18, 64, kernel_size=1, bias=False), self.bn1, _lrelu())
        self.conv2 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1, bias=False), self.bn2, _lrelu())
        self.conv3 = nn.Sequential(nn.Conv2d(64 * 2, 64, kernel_size=1, bias=False), self.bn3, _lrelu())
        self.conv4 = nn.Sequential(nn.Conv2d(64, 64, kernel_size=1, bias=False), self.bn4, _lrelu())
        self.conv5 = nn.Sequential(nn.Conv2d(64 * 2, 64, kernel_size=1, bias=False), self.bn5, _lrelu())
        self.conv6 = nn.Sequential(nn.Conv1d(192, embed_dim, kernel_size=1, bias=False), self.bn6, _lrelu())
        self.conv7 = nn.Sequential(nn.Conv1d(embed_dim + 192, 512, kernel_size=1, bias=False), self.bn7, _lrelu())
        self.conv8 = nn.Sequential(nn.Conv1d(512, 256, kernel_size=1, bias=False), self.bn8, _lrelu())
        self.dp1 = nn.Dropout(dropout)
        self.conv9 = nn.Conv1d(256, feat_dim, kernel_size=1, bias=False)

    def _cba(self, seq, x, maxk=False):
        """conv + BatchNorm + LeakyReLU (+ max over the neighbour dimension): eval mode folds BN + activation (+ max) into one pass."""
        if fused_eval(x, self) and (not maxk or (x.dim() == 4 and x.shape[-1] % 4 == 0 and x.shape[0] * seq[0].out_channels <= 65535)):
            y = seq[0](x)
            scale, shift = folded_bn(seq[1])
            slope = float(seq[2].negative_slope)
            if maxk:
                return ops.affine_act_maxk(y, scale, shift, ops.ACT_LEAKY, slope)
            if y.numel() // (y.shape[0] * y.shape[1]) % 4 == 0 and y.shape[0] * y.shape[1] <= 65535:
                return ops.affine_act(y, scale, shift, ops.ACT_LEAKY, slope)
            return seq[2](seq[1](y))
        y = seq(x)
        return y.max(dim=-1, keepdim=False)[0] if maxk else y

    def _embed(self, x):
        num_points = x.size(2)
        x = get_graph_feature(x, k=self.k, dim9=True)
        x1 = self._cba(self.conv2, self._cba(self.conv1, x), maxk=True)
        x = get_graph_feature(x1, k=self.k)
        x2 = self._cba(self.conv4, self._cba(self.conv3, x), maxk=True)
        x = get_graph_feature(x2, k=self.k)
        x3 = self._cba(self.conv5, x, maxk=True)
        x = self._cba(self.conv6, torch.cat((x1, x2, x3), dim=1))
        x = x.max(dim=-1, keepdim=True)[0].repeat(1, 1, num_points)
        x = torch.cat((x, x1, x2, x3), dim=1)
        x = self._cba(self.conv8, self._cba(self.conv7, x))
        return self.conv9(self.dp1(x))


class DgcnnPcdEmb(_DgcnnTrunk):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.k = args.get("k", 16)
        self.embed_dim = args.get("embed_dim", 1024)
        self.feat_dim = args.get("feat_dim", 128)
        self.dropout = args.get("dropout", 0.1)
        self._build(self.embed_dim, self.feat_dim, self.dropout)

    def forward(self, x):
        return self._embed(x)


class DgcnnMeshEmb(_DgcnnTrunk):
    def __init__(self, args, cls_id, model_points=None):
        super().__init__()
        self.args = args
        self.k = args.get("k", 20)
        self.feat_dim = args.get("feat_dim", 128)
        self.embed_dim = args.get("embed_dim", 1024)
        self.dropout = args.get("dropout", 0.1)
        self.model_pth = args.get("model_pth", "datasets/ycb/ycbv/bop_ycb_kps")
        self.model_id = cls_id
        self.n_mesh_node = args.get("n_mesh_node", 2048)
        self.load_mesh(model_points)
        self._build(self.embed_dim, self.feat_dim, self.dropout)
        self.sys_corr_idx = None

    def load_mesh(self, model_points=None):
        """dgcnn.py:188-202: rows xyz (m), rgb normalised (std .229,.224,.225), normal -> buffer mesh [1,9,M]."""
        if model_points is None:
            model_points = np.load(os.path.join(self.model_pth, "obj_%06d_fps.npy" % self.model_id))
        data = np.asarray(model_points)[: self.n_mesh_node, :9].astype(np.float64)
        data[:, :3] = data[:, :3].astype(np.float32) / 1000.0
        x = data[:, 3:6] / 255.0
        x -= COLOR_MEAN.astype(np.float64)
        x /= COLOR_STD_MESH.astype(np.float64)
        data[:, 3:6] = x
        self.register_buffer("mesh", torch.from_numpy(data.T[np.newaxis, :, :]).float())

    @property
    def xyz(self):
        return self.mesh[0, :3].t()

    def forward(self):
        return self._embed(self.mesh)
