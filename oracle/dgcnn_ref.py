"""ORACLE -- test infrastructure only; never imported by the product package.

Functional torch-CPU restatement (eval mode) of the DGCNN variant:
  /root/reference/models/dgcnn.py:21-27 knn, :30-56 get_graph_feature, :103-136 DgcnnPcdEmb.forward,
  :204-237 DgcnnMeshEmb.forward; /root/reference/models/geoMatch_DGCNN.py:160-180 forward (eval branch).
Driven by a state_dict with the reference's names.  Pinned by tests/golden/dgcnn_eval.npz."""
import torch
import torch.nn.functional as F

from .model_ref import SD, bn, heads_forward


def knn(x, k):
    inner = -2 * torch.matmul(x.transpose(2, 1), x)
    xx = torch.sum(x ** 2, dim=1, keepdim=True)
    pairwise_distance = -xx - inner - xx.transpose(2, 1)
    return pairwise_distance.topk(k=k, dim=-1)[1], pairwise_distance


def get_graph_feature(x, k, dim9=False):
    B, C, n = x.shape
    idx, _ = knn(x[:, :3] if dim9 else x, k)
    idx = (idx + torch.arange(0, B).view(-1, 1, 1) * n).view(-1)
    xt = x.transpose(2, 1).contiguous()
    feature = xt.view(B * n, -1)[idx, :].view(B, n, k, C)
    xt = xt.view(B, n, 1, C).repeat(1, 1, k, 1)
    return torch.cat((feature - xt, xt), dim=3).permute(0, 3, 1, 2).contiguous()


def _cbl(x, s, i):
    w = s["conv%d.0.weight" % i]
    y = F.conv2d(x, w) if w.dim() == 4 else F.conv1d(x, w)
    return F.leaky_relu(bn(y, s.sub("bn%d" % i), 1e-5), 0.2)


def trunk(x, s, k):
    n = x.shape[2]
    x1 = _cbl(_cbl(get_graph_feature(x, k, dim9=True), s, 1), s, 2).max(dim=-1)[0]
    x2 = _cbl(_cbl(get_graph_feature(x1, k), s, 3), s, 4).max(dim=-1)[0]
    x3 = _cbl(get_graph_feature(x2, k), s, 5).max(dim=-1)[0]
    g = _cbl(torch.cat((x1, x2, x3), dim=1), s, 6).max(dim=-1, keepdim=True)[0].repeat(1, 1, n)
    y = _cbl(_cbl(torch.cat((g, x1, x2, x3), dim=1), s, 7), s, 8)
    return F.conv1d(y, s["conv9.weight"])


def trunk_graphs(x, s, k):
    """The three dynamic graphs of one trunk: [(idx [B,n,k], pairwise_distance [B,n,n])] for xyz, x1, x2 (dgcnn.py:108-120)."""
    out = [knn(x[:, :3], k)]
    x1 = _cbl(_cbl(get_graph_feature(x, k, dim9=True), s, 1), s, 2).max(dim=-1)[0]
    out.append(knn(x1, k))
    x2 = _cbl(_cbl(get_graph_feature(x1, k), s, 3), s, 4).max(dim=-1)[0]
    out.append(knn(x2, k))
    return out


def graph_mismatch_not_near_tie(idx, want, dist, tol):
    """Rows whose neighbour SET differs from `want` although the differing candidates are NOT within `tol` (relative to the
    row's k-th score) of each other: 0 means every disagreement is an fp32 near-tie at the k-th place."""
    bad = 0
    a, b = torch.sort(idx, dim=-1)[0], torch.sort(want, dim=-1)[0]
    rows = torch.nonzero((a != b).any(dim=-1))
    for bi, ri in rows.tolist():
        sa, sb = set(idx[bi, ri].tolist()), set(want[bi, ri].tolist())
        d = dist[bi, ri]
        only = torch.tensor(sorted(sa ^ sb))
        spread = (d[only].max() - d[only].min()).abs().item()
        scale = max(1.0, d[idx[bi, ri]].abs().max().item())
        if spread > tol * scale:
            bad += 1
    return bad


def geomatch_dgcnn_forward(sd, cld_rgb_nrm, k_cloud=16, k_mesh=20):
    emb = trunk(cld_rgb_nrm, SD(sd, "pcd_emb."), k_cloud)
    mesh = trunk(sd["model_emb.mesh"], SD(sd, "model_emb."), k_mesh)
    rgbd, seg = heads_forward(sd, emb)
    return dict(seg=seg, mesh=mesh, rgbd=rgbd, emb=emb)
