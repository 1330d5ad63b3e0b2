"""ORACLE -- test infrastructure only; never imported by the product package.

Functional, torch-CPU restatement (eval mode) of the reference network forward, driven by a
state_dict with the REFERENCE's parameter names.  No nn.Module from the product is used: every layer
is F.conv / F.batch_norm on tensors looked up by key, and every neighbour access is written as the
reference writes it (oracle/ops_ref.py).  Line references are to /root/reference.

  ffb6d_forward      models/ffb6d.py:172-285
  dilated_res_block  models/RandLA/RandLANet.py:683-688, building block :700-718, att pooling :747-754
  geomatch_forward   models/geoMatch.py:178-199 (eval branch)
  trunk pieces       models/cnn/extractors.py:36-58,181-200 ; models/cnn/pspnet.py:24-45,108-112

Pinned against the real reference by tests/golden/geomatch_eval.npz and ops_blocks.npz
(tests/test_oracle_model.py).  The SplineCNN mesh branch is NOT pinned (third-party arithmetic, absent):
spline_mesh_forward restates the published operator and is only checked for self-consistency.
"""
import torch
import torch.nn.functional as F

from . import ops_ref


class SD:
    def __init__(self, sd, prefix=""):
        self.sd, self.p = sd, prefix

    def sub(self, name):
        return SD(self.sd, self.p + name + ".")

    def __getitem__(self, k):
        return self.sd[self.p + k]

    def has(self, k):
        return (self.p + k) in self.sd


def bn(x, s, eps):
    return F.batch_norm(x, s["running_mean"], s["running_var"], s["weight"], s["bias"], False, 0.0, eps)


def conv(x, s):
    w = s["weight"]
    b = s["bias"] if s.has("bias") else None
    return F.conv2d(x, w, b) if w.dim() == 4 else F.conv1d(x, w, b)


def pt_conv(x, s, act="relu"):
    """models/pytorch_utils.py:_ConvBase: conv -> normlayer.bn (eps 1e-5) -> ReLU."""
    y = conv(x, s.sub("conv"))
    if s.has("normlayer.bn.weight"):
        y = bn(y, s.sub("normlayer.bn"), 1e-5)
    return F.relu(y) if act == "relu" else y


def rl_conv(x, s, act="lrelu"):
    """models/RandLA/pytorch_utils.py:_ConvBase: conv -> bn.bn (eps 1e-6) -> LeakyReLU(0.2)."""
    y = conv(x, s.sub("conv"))
    if s.has("bn.bn.weight"):
        y = bn(y, s.sub("bn.bn"), 1e-6)
    return F.leaky_relu(y, 0.2) if act == "lrelu" else y


def att_pooling(fset, s):
    att = F.conv2d(fset, s["fc.weight"])
    agg = ops_ref.att_pool_core(att, fset)
    return rl_conv(agg, s.sub("mlp"))


def building_block(xyz, feature, nei, s):
    f_xyz = ops_ref.relative_pos_encoding(xyz, nei)
    f_xyz = rl_conv(f_xyz, s.sub("mlp1"))
    f_nb = ops_ref.group_gather(feature, nei)
    agg = att_pooling(torch.cat([f_nb, f_xyz], dim=1), s.sub("att_pooling_1"))
    f_xyz = rl_conv(f_xyz, s.sub("mlp2"))
    f_nb = ops_ref.group_gather(agg, nei)
    return att_pooling(torch.cat([f_nb, f_xyz], dim=1), s.sub("att_pooling_2"))


def dilated_res_block(feature, xyz, nei, s):
    f = rl_conv(feature, s.sub("mlp1"))
    f = building_block(xyz, f, nei, s.sub("lfa"))
    f = rl_conv(f, s.sub("mlp2"), act=None)
    sc = rl_conv(feature, s.sub("shortcut"), act=None)
    return F.leaky_relu(f + sc, negative_slope=0.2)


def basic_block(x, s, stride):
    out = F.relu(bn(F.conv2d(x, s["conv1.weight"], None, stride, 1), s.sub("bn1"), 1e-5))
    out = bn(F.conv2d(out, s["conv2.weight"], None, 1, 1), s.sub("bn2"), 1e-5)
    res = x
    if s.has("downsample.0.weight"):
        res = bn(F.conv2d(x, s["downsample.0.weight"], None, stride), s.sub("downsample.1"), 1e-5)
    return F.relu(out + res)


def res_layer(x, s, stride):
    x = basic_block(x, s.sub("0"), stride)
    return basic_block(x, s.sub("1"), 1)


def psp_module(f, s):
    h, w = f.shape[2], f.shape[3]
    pri = []
    for i, size in enumerate((1, 2, 3, 6)):
        p = F.adaptive_avg_pool2d(f, (size, size))
        p = F.conv2d(p, s["stages.%d.1.weight" % i])
        pri.append(F.interpolate(p, size=(h, w), mode="bilinear", align_corners=True))
    pri.append(f)
    return F.relu(F.conv2d(torch.cat(pri, 1), s["bottleneck.weight"], s["bottleneck.bias"]))


def psp_upsample(x, s):
    x = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    x = F.conv2d(x, s["conv.1.weight"], s["conv.1.bias"], padding=1)
    x = bn(x, s.sub("conv.2"), 1e-5)
    return F.prelu(x, s["conv.3.weight"])


def final_stage(x, s):
    return F.log_softmax(F.conv2d(x, s["0.weight"], s["0.bias"]), dim=1)   # nn.LogSoftmax() implicit dim -> 1 for 4-D


def ffb6d_forward(sd, inputs, prefix="pcd_emb."):
    s = SD(sd, prefix)
    L = lambda k: inputs[k].long()
    x = F.conv2d(inputs["rgb"], s["cnn_pre_stages.0.weight"], None, 2, 3)
    x = F.relu(bn(x, s.sub("cnn_pre_stages.1"), 1e-5))
    rgb_emb = F.max_pool2d(x, 3, 2, 1)
    p_emb = rl_conv(inputs["cld_rgb_nrm"], s.sub("rndla_pre_stages")).unsqueeze(3)

    ds_emb = []
    for i in range(4):
        cs = s.sub("cnn_ds_stages.%d" % i)
        if i == 0:
            rgb_emb0 = res_layer(rgb_emb, cs, 1)
        elif i == 1:
            rgb_emb0 = res_layer(rgb_emb, cs, 2)
        elif i == 2:
            rgb_emb0 = res_layer(res_layer(rgb_emb, cs.sub("0"), 1), cs.sub("1"), 1)
        else:
            rgb_emb0 = psp_module(rgb_emb, cs.sub("0"))                 # Dropout2d is identity in eval
        bs, c, hr, wr = rgb_emb0.shape
        f_enc = dilated_res_block(p_emb, inputs["cld_xyz%d" % i], L("cld_nei_idx%d" % i), s.sub("rndla_ds_stages.%d" % i))
        p_emb0 = ops_ref.random_sample(f_enc, L("cld_sub_idx%d" % i))
        if i == 0:
            ds_emb.append(f_enc)
        p2r = pt_conv(p_emb0, s.sub("ds_fuse_p2r_pre_layers.%d" % i))
        p2r = ops_ref.nearest_interpolation(p2r, L("p2r_ds_nei_idx%d" % i)).view(bs, -1, hr, wr)
        rgb_emb = pt_conv(torch.cat((rgb_emb0, p2r), dim=1), s.sub("ds_fuse_p2r_fuse_layers.%d" % i))
        r2p = ops_ref.random_sample(rgb_emb0.reshape(bs, c, hr * wr, 1), L("r2p_ds_nei_idx%d" % i)).view(bs, c, -1, 1)
        r2p = pt_conv(r2p, s.sub("ds_fuse_r2p_pre_layers.%d" % i))
        p_emb = pt_conv(torch.cat((p_emb0, r2p), dim=1), s.sub("ds_fuse_r2p_fuse_layers.%d" % i))
        ds_emb.append(p_emb)

    n_up = 4
    for i in range(n_up - 1):
        us = s.sub("cnn_up_stages.%d" % i)
        rgb_emb0 = psp_upsample(rgb_emb, us.sub("0")) if i < 2 else final_stage(rgb_emb, us.sub("0"))
        bs, c, hr, wr = rgb_emb0.shape
        f_int = ops_ref.nearest_interpolation(p_emb, L("cld_interp_idx%d" % (n_up - i - 1)))
        p_emb0 = rl_conv(torch.cat([ds_emb[-i - 2], f_int], dim=1), s.sub("rndla_up_stages.%d" % i))
        p2r = pt_conv(p_emb0, s.sub("up_fuse_p2r_pre_layers.%d" % i))
        p2r = ops_ref.nearest_interpolation(p2r, L("p2r_up_nei_idx%d" % i)).view(bs, -1, hr, wr)
        rgb_emb = pt_conv(torch.cat((rgb_emb0, p2r), dim=1), s.sub("up_fuse_p2r_fuse_layers.%d" % i))
        r2p = ops_ref.random_sample(rgb_emb0.reshape(bs, c, hr * wr), L("r2p_up_nei_idx%d" % i)).view(bs, c, -1, 1)
        r2p = pt_conv(r2p, s.sub("up_fuse_r2p_pre_layers.%d" % i))
        p_emb = pt_conv(torch.cat((p_emb0, r2p), dim=1), s.sub("up_fuse_r2p_fuse_layers.%d" % i))

    us = s.sub("cnn_up_stages.3")
    rgb_emb = final_stage(psp_upsample(rgb_emb, us.sub("0")), us.sub("1"))
    f_int = ops_ref.nearest_interpolation(p_emb, L("cld_interp_idx0"))
    p_emb = rl_conv(torch.cat([ds_emb[0], f_int], dim=1), s.sub("rndla_up_stages.3")).squeeze(-1)
    bs, di = rgb_emb.shape[:2]
    rgb_c = torch.gather(rgb_emb.view(bs, di, -1), 2, L("choose").repeat(1, di, 1)).contiguous()
    return torch.cat([rgb_c, p_emb], dim=1)


def heads_forward(sd, rgbd_emb):
    """geoMatch.py:180-183 -> (rgbd_features, seg_features)."""
    s = SD(sd)
    x = rgbd_emb
    for i in range(3):
        x = pt_conv(x, s.sub("feature_encoding_layer.%d" % i))
    rgbd_features = pt_conv(x, s.sub("feature_encoding_layer.3"), act=None)
    normalized = pt_conv(rgbd_features, s.sub("normalize_feature_layer"))
    y = rgbd_emb + normalized
    for i in range(3):
        y = pt_conv(y, s.sub("seg_layer.%d" % i))
    seg = pt_conv(y, s.sub("seg_layer.3"), act=None)
    return rgbd_features, seg


def geomatch_forward(sd, inputs, mesh_features):
    emb = ffb6d_forward(sd, inputs)
    rgbd, seg = heads_forward(sd, emb)
    return dict(seg=seg, mesh=mesh_features.unsqueeze(0), rgbd=rgbd, emb=emb)


# ------------------------------------------------------------------------------ SplineCNN (parity unpinned)
def spline_basis(pseudo, ks=5):
    """torch_spline_conv basis (degree 1, open): pseudo [E,3] -> basis [E,8], weight index [E,8]."""
    E = pseudo.shape[0]
    v = pseudo * (ks - 1)
    fl = torch.floor(v)
    fr = v - fl
    fl = fl.long()
    basis = torch.ones(E, 8)
    wi = torch.zeros(E, 8, dtype=torch.long)
    for s_ in range(8):
        off = 1
        for d in range(3):
            kd = (s_ >> d) & 1
            wi[:, s_] += ((fl[:, d] + kd) % ks) * off
            off *= ks
            basis[:, s_] *= fr[:, d] if kd else (1 - fr[:, d])
    return basis, wi


def spline_conv(x, edge_index, edge_attr, weight, root_w, bias):
    """SplineConv (mean aggregation at edge_index[1] of messages from edge_index[0]):
    msg_e = sum_s basis[e,s] * x[src_e] @ W[wi[e,s]].  Edges are visited grouped by kernel index so that no
    [E, in, out] weight gather is materialised (M = 8192 would need 2 GB per corner)."""
    M = x.shape[0]
    basis, wi = spline_basis(edge_attr)
    xj = x[edge_index[0]]
    msg = torch.zeros(edge_index.shape[1], weight.shape[2])
    for s_ in range(8):
        order = torch.argsort(wi[:, s_], stable=True)
        counts = torch.bincount(wi[:, s_], minlength=weight.shape[0]).tolist()
        pos = 0
        for kidx, cnt in enumerate(counts):
            if cnt:
                e = order[pos:pos + cnt]
                msg[e] += basis[e, s_:s_ + 1] * (xj[e] @ weight[kidx])
                pos += cnt
    out = torch.zeros(M, weight.shape[2]).index_add_(0, edge_index[1], msg)
    deg = torch.bincount(edge_index[1], minlength=M).clamp(min=1).unsqueeze(1).float()
    return out / deg + x @ root_w.t() + bias


def mesh_graph(pos, k=4):
    """KNNGraph(k, loop=False) + Cartesian(norm=True): brute force on CPU."""
    from . import knn as oknn
    idx = torch.from_numpy(oknn.knn_batch(pos.numpy()[None], pos.numpy()[None], k + 1)[0])
    M = pos.shape[0]
    centre = torch.arange(M).unsqueeze(1).expand(M, k + 1)
    keep = idx != centre
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)[:, :k]
    nbr = torch.gather(idx, 1, order)
    row, col = nbr.reshape(-1), centre[:, :k].reshape(-1)
    cart = pos[row] - pos[col]
    cart = cart / (2 * cart.abs().max()) + 0.5
    return torch.stack([row, col]), cart


def spline_mesh_forward(sd, prefix="model_emb."):
    s = SD(sd, prefix)
    x = s["mesh_graph_x"]
    ei, ea = s["mesh_graph_edge_index"], s["mesh_graph_edge_attr"]
    feats = [x]
    for i in range(3):
        c = s.sub("mesh_convs.%d" % i)
        feats.append(F.relu(spline_conv(feats[-1], ei, ea, c["weight"], c["lin.weight"], c["bias"])))
    out = torch.cat(feats, dim=-1)
    out = out @ s["mesh_final.weight"].t() + s["mesh_final.bias"]
    return out.transpose(0, 1)
