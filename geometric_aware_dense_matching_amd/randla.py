"""RandLA-Net local feature aggregation blocks on the HIP gather / pooling kernels.

Mirrors (names, shapes, semantics) /root/reference/models/RandLA/RandLANet.py:
  Dilated_res_block :674-688   Building_block :691-738   Att_pooling :741-754
Only these blocks are used by the geoMatch path (models/ffb6d.py:28,40,90).

What changes versus the reference is HOW the neighbour access runs: instead of
permute -> index.repeat over channels -> torch.gather -> permute -> contiguous (five passes and an
int64 index of B*n*K*C elements), one HIP kernel reads the int32 index row once and writes the
channel-major [B,C,n,K] tensor directly (ops.group_gather), the 10-channel relative position
encoding is one kernel (ops.rel_pos_enc), and softmax-over-K * feature -> sum is one kernel
(ops.att_pool).  Features stay [B,C,n,1] / [B,C,n,K] channel-major exactly as in the reference,
so the 1x1 convolutions are unchanged MIOpen/rocBLAS GEMMs.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, settings
from .layers import folded_bn, fused_eval, rl_conv2d



class AttPooling(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.fc = nn.Conv2d(d_in, d_in, (1, 1), bias=False)
        self.mlp = rl_conv2d(d_in, d_out, bn=True)

    def forward(self, feature_set):                       # [B,C,n,K]
        if self.training and ops.conv1x1_train_supported(self.fc, feature_set):
            att_activation = ops.conv1x1_train(self.fc, feature_set)
        else:
            att_activation = self.fc(feature_set)
        f_agg = ops.att_pool(att_activation, feature_set)  # [B,C,n]
        return self.mlp(f_agg.unsqueeze(3))


class BuildingBlock(nn.Module):
    def __init__(self, d_out):
        super().__init__()
        self.mlp1 = rl_conv2d(10, d_out // 2, bn=True)
        self.att_pooling_1 = AttPooling(d_out, d_out // 2)
        self.mlp2 = rl_conv2d(d_out // 2, d_out // 2, bn=True)
        self.att_pooling_2 = AttPooling(d_out, d_out)

    def _fused_weights(self):
        """Transposed weights + folded BatchNorms of both stages, cached until any parameter changes."""
        convs = (self.mlp1, self.mlp2, self.att_pooling_1.mlp, self.att_pooling_2.mlp)
        deps = [m.conv.weight for m in convs] + [self.att_pooling_1.fc.weight, self.att_pooling_2.fc.weight]
        for m in convs:
            deps += [m.bn.bn.weight, m.bn.bn.bias, m.bn.bn.running_mean, m.bn.bn.running_var]
        key = tuple((t._version, t.data_ptr()) for t in deps)
        cache = self.__dict__.get("_gdm_lfa")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                def wt(conv):
                    return conv.weight.view(conv.weight.shape[0], -1).t().contiguous()
                w = dict(w1t=wt(self.mlp1.conv), w2t=wt(self.mlp2.conv), wf1=wt(self.att_pooling_1.fc), wf2=wt(self.att_pooling_2.fc),
                         wm1=wt(self.att_pooling_1.mlp.conv), wm2=wt(self.att_pooling_2.mlp.conv))
                w["s1"], w["b1"] = folded_bn(self.mlp1.bn.bn)
                w["s2"], w["b2"] = folded_bn(self.mlp2.bn.bn)
                w["sm1"], w["bm1"] = folded_bn(self.att_pooling_1.mlp.bn.bn)
                w["sm2"], w["bm2"] = folded_bn(self.att_pooling_2.mlp.bn.bn)
            cache = (key, w)
            self.__dict__["_gdm_lfa"] = cache
        return cache[1]

    def _fusable(self, feature, neigh_idx):
        acts = (self.mlp1, self.mlp2, self.att_pooling_1.mlp, self.att_pooling_2.mlp)
        return (settings.USE_FUSED_LFA and fused_eval(feature, self) and ops.lfa_supported(2 * feature.shape[1], neigh_idx.shape[-1])
                and all(isinstance(getattr(m, "activation", None), nn.LeakyReLU) and m.activation.negative_slope == 0.2
                        and m.conv.bias is None for m in acts))

    def forward(self, xyz, feature, neigh_idx):            # feature [B,C,n,1]
        if self._fusable(feature, neigh_idx):
            # both attentive-pooling stages as one launch each: nothing of size n*K reaches HBM
            w = self._fused_weights()
            B, H, n = feature.shape[0], feature.shape[1], feature.shape[2]
            agg = ops.lfa_stage(xyz, neigh_idx, feature.reshape(B, H, n), w["w1t"], w["s1"], w["b1"], None, None, None,
                                w["wf1"], w["wm1"], w["sm1"], w["bm1"])
            out = ops.lfa_stage(xyz, neigh_idx, agg, w["w1t"], w["s1"], w["b1"], w["w2t"], w["s2"], w["b2"],
                                w["wf2"], w["wm2"], w["sm2"], w["bm2"])
            return out.unsqueeze(3)
        f_xyz = ops.rel_pos_enc(xyz, neigh_idx)             # [B,10,n,K]
        f_xyz = self.mlp1(f_xyz)
        f_neighbours = ops.group_gather(feature, neigh_idx)  # [B,C,n,K]
        f_pc_agg = self.att_pooling_1(torch.cat([f_neighbours, f_xyz], dim=1))
        f_xyz = self.mlp2(f_xyz)
        f_neighbours = ops.group_gather(f_pc_agg, neigh_idx)
        return self.att_pooling_2(torch.cat([f_neighbours, f_xyz], dim=1))


class DilatedResBlock(nn.Module):
    def __init__(self, d_in, d_out):
        super().__init__()
        self.mlp1 = rl_conv2d(d_in, d_out // 2, bn=True)
        self.lfa = BuildingBlock(d_out)
        self.mlp2 = rl_conv2d(d_out, d_out * 2, bn=True, activation=None)
        self.shortcut = rl_conv2d(d_in, d_out * 2, bn=True, activation=None)

    def forward(self, feature, xyz, neigh_idx, f_pc=None):
        """f_pc: mlp1(feature) when the caller already has it (ffb6d: the stem layer and this block's mlp1 as one launch)."""
        if f_pc is None:
            f_pc = self.mlp1(feature)
        f_pc = self.lfa(xyz, f_pc, neigh_idx)
        if fused_eval(feature, self) and settings.USE_POINTWISE:
            # lrelu(bn(mlp2(f_pc)) + bn(shortcut(feature))): both 1x1 layers, both folded BatchNorms, the sum and the activation in ONE launch
            return self.mlp2.forward_segs([f_pc], res=(self.shortcut, feature), act=(ops.ACT_LEAKY, 0.2))
        if fused_eval(feature, self):
            sa, ba = folded_bn(self.mlp2.bn.bn)
            sr, br = folded_bn(self.shortcut.bn.bn)
            return ops.affine_act(self.mlp2.conv(f_pc), sa, ba, ops.ACT_LEAKY, 0.2, res=self.shortcut.conv(feature),
                                  res_scale=sr, res_shift=br)
        f_pc = self.mlp2(f_pc)
        shortcut = self.shortcut(feature)
        return F.leaky_relu(f_pc + shortcut, negative_slope=0.2)
