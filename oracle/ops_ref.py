"""ORACLE -- test infrastructure only; never imported by the product package.

Plain torch-CPU restatements of the reference's gather / pooling / matching op chains, one
function per chain, each citing the lines it follows under /root/reference.  They are
deliberately written the way the reference writes them (materialised repeated indices,
torch.gather, separate softmax / multiply / sum), so they also serve as the "reference CPU
path" timed by bench.py's cpu_baseline leg.

Pinned against the reference's own functions (imported in this container) by
tests/golden/make_golden.py -> tests/golden/ops_*.npz, checked in tests/test_oracle_ops.py.
"""
import torch
import torch.nn.functional as F


def random_sample(feature, pool_idx):
    """models/ffb6d.py:128-146. feature [B,C,n(,1)], pool_idx int64 [B,n',K] -> [B,C,n',1]."""
    if feature.dim() > 3:
        feature = feature.squeeze(dim=3)
    num_neigh = pool_idx.shape[-1]
    d = feature.shape[1]
    batch_size = pool_idx.shape[0]
    pool_idx = pool_idx.reshape(batch_size, -1)
    pool_features = torch.gather(feature, 2, pool_idx.unsqueeze(1).repeat(1, feature.shape[1], 1)).contiguous()
    pool_features = pool_features.reshape(batch_size, d, -1, num_neigh)
    return pool_features.max(dim=3, keepdim=True)[0]


def nearest_interpolation(feature, interp_idx):
    """models/ffb6d.py:148-163. feature [B,C,n,1], interp_idx int64 [B,n',1] -> [B,C,n',1]."""
    feature = feature.squeeze(dim=3)
    batch_size = interp_idx.shape[0]
    up_num_points = interp_idx.shape[1]
    interp_idx = interp_idx.reshape(batch_size, up_num_points)
    out = torch.gather(feature, 2, interp_idx.unsqueeze(1).repeat(1, feature.shape[1], 1)).contiguous()
    return out.unsqueeze(3)


def gather_neighbour(pc, neighbor_idx):
    """models/RandLA/RandLANet.py:729-738. pc [B,n,d], idx int64 [B,n',K] -> [B,n',K,d]."""
    batch_size = pc.shape[0]
    d = pc.shape[2]
    index_input = neighbor_idx.reshape(batch_size, -1)
    features = torch.gather(pc, 1, index_input.unsqueeze(-1).repeat(1, 1, pc.shape[2])).contiguous()
    return features.reshape(batch_size, neighbor_idx.shape[1], neighbor_idx.shape[-1], d)


def group_gather(feature, neighbor_idx):
    """gather_neighbour on channel-major features + the permute of RandLANet.py:704-707:
    feature [B,C,n(,1)] -> [B,C,n',K]."""
    if feature.dim() > 3:
        feature = feature.squeeze(-1)
    f = gather_neighbour(feature.permute(0, 2, 1).contiguous(), neighbor_idx)
    return f.permute(0, 3, 1, 2).contiguous()


def relative_pos_encoding(xyz, neigh_idx):
    """models/RandLA/RandLANet.py:720-727 (+ permute :701-702) -> [B,10,n,K]."""
    neighbor_xyz = gather_neighbour(xyz, neigh_idx)
    xyz_tile = xyz.unsqueeze(2).repeat(1, 1, neigh_idx.shape[-1], 1)
    relative_xyz = xyz_tile - neighbor_xyz
    relative_dis = torch.sqrt(torch.sum(torch.pow(relative_xyz, 2), dim=-1, keepdim=True))
    relative_feature = torch.cat([relative_dis, relative_xyz, xyz_tile, neighbor_xyz], dim=-1)
    return relative_feature.permute(0, 3, 1, 2).contiguous()


def att_pool_core(att_activation, feature_set):
    """models/RandLA/RandLANet.py:749-752 without the two convolutions: [B,C,n,K] x2 -> [B,C,n,1]."""
    att_scores = F.softmax(att_activation, dim=3)
    f_agg = feature_set * att_scores
    return torch.sum(f_agg, dim=3, keepdim=True)


def match_argmax(rgbd_features, mesh_features, cls_msk=None):
    """evaluator.py:81-93 for one crop. rgbd_features [D,N], mesh_features [D,M], optional bool mask [N]
    -> (max_th [n_sel], obj_pts_idx int64 [n_sel], obj_pts_sim [n_sel,M])."""
    rgbd = rgbd_features.transpose(0, 1)
    if cls_msk is not None:
        rgbd = rgbd[cls_msk]
    sel = F.normalize(rgbd, p=2, dim=1)
    mesh = F.normalize(mesh_features, p=2, dim=0)
    sim = torch.matmul(sel, mesh)
    max_th, idx = torch.max(sim, dim=1)
    return max_th, idx, sim


def seg_mask(seg_features):
    """evaluator.py:79-83: seg [2,N] -> bool [N]."""
    return torch.argmax(seg_features, dim=0) == 1


def pdist(A, B):
    """utils/basic_utils.py:86-89 (L2 branch)."""
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(D2 + 1e-7)
