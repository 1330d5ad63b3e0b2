"""FFB6DEmb: bidirectional fusion of the image branch and the RandLA point branch.

Mirrors /root/reference/models/ffb6d.py:9-285 (constructor layout and parameter names identical,
so `pcd_emb.*` checkpoint keys load unchanged; forward follows :172-285 stage by stage).

HIP operators replace the reference's gather chains:
  random_sample (:128-146)          -> ops.gather_max   (index read once, max over K in registers)
  nearest_interpolation (:148-163)  -> ops.gather_nn
  choose gather (:278-281)          -> ops.gather_nn
Neighbour indices are consumed as int32 (int64 accepted and narrowed).
"""
import torch
import torch.nn as nn

from . import ops, settings
from .cnn import PSPNet, bn_act
from .layers import act_code, cached_gemm_weight, folded_bn, fused_eval, pt_conv2d, rl_conv1d, rl_conv2d
from .randla import DilatedResBlock


class FFB6DEmb(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        cnn = PSPNet()
        d_outs = list(cfg.d_out)
        n_layers = cfg.num_layers

        self.cnn_pre_stages = nn.Sequential(cnn.feats.conv1, cnn.feats.bn1, cnn.feats.relu, cnn.feats.maxpool)
        self.rndla_pre_stages = rl_conv1d(cfg.in_c, 8, bn=True)                  # RandLANet.py:19 fc0

        self.cnn_ds_stages = nn.ModuleList([
            cnn.feats.layer1,
            cnn.feats.layer2,
            nn.Sequential(cnn.feats.layer3, cnn.feats.layer4),
            nn.Sequential(cnn.psp, cnn.drop_1),
        ])
        self.ds_sr = [4, 8, 8, 8]

        blocks = nn.ModuleList()                                                 # RandLANet.py:21-26
        d_in = 8
        for i in range(n_layers):
            blocks.append(DilatedResBlock(d_in, d_outs[i]))
            d_in = 2 * d_outs[i]
        self.rndla_ds_stages = blocks

        self.ds_rgb_oc = [64, 128, 512, 1024]
        self.ds_rndla_oc = [c * 2 for c in d_outs]
        self.ds_fuse_r2p_pre_layers = nn.ModuleList()
        self.ds_fuse_r2p_fuse_layers = nn.ModuleList()
        self.ds_fuse_p2r_pre_layers = nn.ModuleList()
        self.ds_fuse_p2r_fuse_layers = nn.ModuleList()
        for i in range(4):
            self.ds_fuse_r2p_pre_layers.append(pt_conv2d(self.ds_rgb_oc[i], self.ds_rndla_oc[i], bn=True))
            self.ds_fuse_r2p_fuse_layers.append(pt_conv2d(self.ds_rndla_oc[i] * 2, self.ds_rndla_oc[i], bn=True))
            self.ds_fuse_p2r_pre_layers.append(pt_conv2d(self.ds_rndla_oc[i], self.ds_rgb_oc[i], bn=True))
            self.ds_fuse_p2r_fuse_layers.append(pt_conv2d(self.ds_rgb_oc[i] * 2, self.ds_rgb_oc[i], bn=True))

        self.cnn_up_stages = nn.ModuleList([
            nn.Sequential(cnn.up_1, cnn.drop_2),
            nn.Sequential(cnn.up_2, cnn.drop_2),
            nn.Sequential(cnn.final),
            nn.Sequential(cnn.up_3, cnn.final),              # `final` is shared, as in the reference (:79-80)
        ])
        self.up_rgb_oc = [256, 64, 64]
        self.up_rndla_oc = []
        for j in range(n_layers):
            self.up_rndla_oc.append(self.ds_rndla_oc[-j - 2] if j < 3 else self.ds_rndla_oc[0])

        dec = nn.ModuleList()                                                    # RandLANet.py:28-39 decoder_blocks
        d_out = 2 * d_outs[-1]
        for j in range(n_layers):
            if j < 3:
                d_in = d_out + 2 * d_outs[-j - 2]
                d_out = 2 * d_outs[-j - 2]
            else:
                d_in = 4 * d_outs[-4]
                d_out = 2 * d_outs[-4]
            dec.append(rl_conv2d(d_in, d_out, bn=True))
        self.rndla_up_stages = dec

        self.up_fuse_r2p_pre_layers = nn.ModuleList()
        self.up_fuse_r2p_fuse_layers = nn.ModuleList()
        self.up_fuse_p2r_pre_layers = nn.ModuleList()
        self.up_fuse_p2r_fuse_layers = nn.ModuleList()
        for i in range(3):
            self.up_fuse_r2p_pre_layers.append(pt_conv2d(self.up_rgb_oc[i], self.up_rndla_oc[i], bn=True))
            self.up_fuse_r2p_fuse_layers.append(pt_conv2d(self.up_rndla_oc[i] * 2, self.up_rndla_oc[i], bn=True))
            self.up_fuse_p2r_pre_layers.append(pt_conv2d(self.up_rndla_oc[i], self.up_rgb_oc[i], bn=True))
            self.up_fuse_p2r_fuse_layers.append(pt_conv2d(self.up_rgb_oc[i] * 2, self.up_rgb_oc[i], bn=True))

    # same names and contracts as the reference's static methods (ffb6d.py:128-163)
    @staticmethod
    def random_sample(feature, pool_idx):
        return ops.gather_max(feature, pool_idx).unsqueeze(3)

    @staticmethod
    def nearest_interpolation(feature, interp_idx):
        return ops.gather_nn(feature, interp_idx).unsqueeze(3)

    @staticmethod
    def _split_fuse_weight(layer, c_first):
        """1x1 fuse conv over cat(a, b): W = [W_a | W_b] split at channel c_first (contiguous copies, cached)."""
        w = layer.conv.weight
        key = (w._version, w.data_ptr(), c_first)
        cache = layer.__dict__.get("_gdm_split")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                w2 = w.view(w.shape[0], -1)
                cache = (key, w2[:, :c_first].contiguous(), w2[:, c_first:].contiguous())
            layer.__dict__["_gdm_split"] = cache
        return cache[1], cache[2]

    @staticmethod
    def _fuse_weight_t(layer, wa, tag="a"):
        """wa transposed ([ci][co], contiguous), cached beside the split weights."""
        slot = "_gdm_w%s_t" % tag
        cache = layer.__dict__.get(slot)
        if cache is None or cache[0] is not wa:
            cache = (wa, wa.t().contiguous())
            layer.__dict__[slot] = cache
        return cache[1]

    def _p2r_point_term(self, pre_layer, fuse_layer, c, p_emb0):
        """The point half of the p2r fusion, W_b . pre(p_emb0), at the points -- exactly what _p2r_fuse computes first (same branch
        conditions); the two-stream pipeline forms it on the POINT stream so that the image stream only waits for the finished term.
        None when _p2r_fuse would not take a path with a separate point term."""
        if not fused_eval(p_emb0, self) or act_code(getattr(fuse_layer, "activation", None)) is None or not settings.USE_POINTWISE:
            return None
        bs = p_emb0.shape[0]
        wa, wb = self._split_fuse_weight(fuse_layer, c)
        pp = pre_layer(p_emb0).reshape(bs, wb.shape[1], -1)
        if c == 64 and wa.shape[0] == 64 and settings.USE_MFMA_GEMM:
            return ops.pointwise([pp], self._fuse_weight_t(fuse_layer, wb, "b"), point_major=True)       # [B, n', 64]
        return ops.pointwise([pp], self._fuse_weight_t(fuse_layer, wb, "b"))                             # [B, Cout, n']

    def _p2r_fuse(self, pre_layer, fuse_layer, rgb_emb0, p_emb0, idx, pixel_major=False, point_term=None, want_packed=False):
        """fuse(cat(rgb_emb0, nearest_interp(pre(p_emb0)))) (ffb6d.py:216-222,252-258).  Eval: the point half of the
        1x1 fuse convolution runs at the points (a 1x1 conv commutes with the gather), the pixel half is a GEMM with half
        the K, and gather + add + BN + ReLU is one HIP launch; no concat, no full-resolution point features."""
        bs, c, hr, wr = rgb_emb0.shape
        if fused_eval(rgb_emb0, self):
            code = act_code(getattr(fuse_layer, "activation", None))
            if code is not None:
                wa, wb = self._split_fuse_weight(fuse_layer, c)
                if c == 64 and wa.shape[0] == 64 and settings.USE_MFMA_GEMM:
                    # K = 64: channel mix on the matrix cores (split-bf16 x3) + gather + add + BN + ReLU in ONE pass over the pixels,
                    # bound by the map's read + write; the point term is formed point-major ([B, n', 64]: one contiguous row per
                    # gathered point) by the same library GEMM with its operands swapped
                    scale, shift = folded_bn(fuse_layer.normlayer.bn)
                    if point_term is not None:
                        t_pm = point_term
                    else:
                        pp = pre_layer(p_emb0).reshape(bs, wb.shape[1], -1)
                        if settings.USE_POINTWISE:
                            t_pm = ops.pointwise([pp], self._fuse_weight_t(fuse_layer, wb, "b"), point_major=True)   # [B, n', 64]
                        else:
                            t_pm = torch.matmul(pp.transpose(1, 2), wb.t())
                    cache = fuse_layer.__dict__.get("_gdm_wa_pk")
                    if cache is None or cache[0] is not wa:
                        cache = (wa, ops.pack_rows64(wa))
                        fuse_layer.__dict__["_gdm_wa_pk"] = cache
                    y = ops.conv64_gather_add_act_mfma(rgb_emb0.reshape(bs, c, hr * wr), cache[1], t_pm, idx.reshape(bs, -1), scale, shift,
                                                       code[0], code[1], pixel_major=pixel_major, t_point_major=True,
                                                       hw=(hr, wr) if (want_packed and settings.USE_PACKED_PRODUCERS and not pixel_major) else None)
                    if pixel_major:
                        return y
                    out = y.view(bs, -1, hr, wr)
                    if getattr(y, "_gdm_packed", None) is not None:
                        out._gdm_packed = y._gdm_packed         # the next stage's first convolution reads this: no pack launch
                    return out
                if point_term is not None:
                    t = point_term
                else:
                    pp = pre_layer(p_emb0).reshape(bs, wb.shape[1], -1)
                    if settings.USE_POINTWISE:
                        t = ops.pointwise([pp], self._fuse_weight_t(fuse_layer, wb, "b"))   # [B,Cout,n'] at the points
                    else:
                        t = ops.wx(wb, pp)
                if c == 64 and wa.shape[0] == 64:
                    # K = 64: GEMM + gather + add + BN + ReLU in ONE pass over the pixels (exact fp32 FMAs)
                    scale, shift = folded_bn(fuse_layer.normlayer.bn)
                    y = ops.conv1x1_gather_add_act(rgb_emb0.reshape(bs, c, hr * wr), self._fuse_weight_t(fuse_layer, wa), t,
                                                   idx.reshape(bs, -1), scale, shift, code[0], code[1], pixel_major=pixel_major)
                    return y if pixel_major else y.view(bs, -1, hr, wr)  # pixel-major: [B, H*W, 64] for _final_at_choose
                if pixel_major:
                    raise RuntimeError("pixel-major fusion output is the 64-channel kernel's; _sparse_final_ok() guards the caller")
                if settings.USE_MFMA_GEMM and ops.gemm_supported(c, wa.shape[0], hr * wr):
                    wpk, co = cached_gemm_weight(fuse_layer, "wa", wa, (fuse_layer.conv.weight,))
                    x = ops.gemm_bf16x3_map(rgb_emb0, wpk, co).view(bs, co, hr * wr)   # split-bf16 MFMA; reads the stage's packed output
                else:
                    x = ops.wx(wa, rgb_emb0.reshape(bs, c, hr * wr))                    # [B,Cout,HW]
                scale, shift = folded_bn(fuse_layer.normlayer.bn)
                y, ypk = ops.gather_add_affine_act(x, t, idx.reshape(bs, -1), scale, shift, code[0], code[1], hw=(hr, wr))
                y = y.view(bs, -1, hr, wr)
                if ypk is not None:
                    y._gdm_packed = ypk             # the next image stage's first convolution / GEMM reads this: no pack launch
                return y
        if pixel_major:
            raise RuntimeError("pixel-major fusion output is the eval kernel's; _sparse_final_ok() guards the caller")
        p2r_emb = self.nearest_interpolation(pre_layer(p_emb0), idx).view(bs, -1, hr, wr)
        return fuse_layer(torch.cat((rgb_emb0, p2r_emb), dim=1))

    def forward(self, inputs, end_points=None, parts=False):
        """-> f32[B,128,N] (ffb6d.py:285: cat of the 64 image channels at the chosen pixels and the 64 point channels); parts=True
        returns the two halves un-concatenated, for a consumer that reads them in place (the fused per-point heads)."""
        if fused_eval(inputs["rgb"], self):
            pre = self.cnn_pre_stages                                         # conv1, bn1, relu, maxpool
            s0, b0 = folded_bn(pre[1])
            mp = pre[3]
            conv = pre[0]
            plain_pool = (isinstance(mp, nn.MaxPool2d) and mp.kernel_size in (3, (3, 3)) and mp.stride in (2, (2, 2)) and mp.padding in (1, (1, 1))
                          and mp.dilation in (1, (1, 1)) and not mp.ceil_mode and isinstance(pre[2], nn.ReLU))
            if (settings.USE_OWN_STEM and plain_pool and tuple(conv.weight.shape) == (64, 3, 7, 7) and conv.bias is None
                    and tuple(conv.stride) == (2, 2) and tuple(conv.padding) == (3, 3) and tuple(conv.dilation) == (1, 1)
                    and inputs["rgb"].shape[0] <= 65535):
                # the whole stem in one own launch (split-bf16 MFMA implicit GEMM, pooled in LDS): no library kernel is left in the step
                key = (conv.weight._version, conv.weight.data_ptr())
                cache = conv.__dict__.get("_gdm_stem_pk")
                if cache is None or cache[0] != key:
                    cache = (key, ops.stem_pack_weight(conv.weight))
                    conv.__dict__["_gdm_stem_pk"] = cache
                rgb_emb = ops.stem(inputs["rgb"], cache[1], s0, b0)
            else:
                y0 = conv(inputs["rgb"])
                if plain_pool and y0.shape[0] * y0.shape[1] <= 65535:
                    rgb_emb = ops.affine_relu_maxpool(y0, s0, b0)          # BN + ReLU + max-pool: one pass over the stem's map
                else:
                    rgb_emb = mp(ops.affine_act(y0, s0, b0, ops.ACT_RELU))
        else:
            pre = self.cnn_pre_stages
            rgb_emb = pre[3](bn_act(pre[1], pre[0](inputs["rgb"]), pre[2]))
        # Inference, settings.USE_SIDE_STREAMS (off by default): the point branch of every encoder stage (RandLA block + pooling) is
        # forked onto side stream 0 beside the image branch's convolutions; they meet at the fusion.  An overlapped neighbour-pyramid
        # build (pyramid.build_pyramid(..., overlap=True)) is waited for by EACH consuming stream before its first index use.
        from . import pyramid as _pyr
        overlap = settings.USE_SIDE_STREAMS and "point" in settings.SIDE_PARTS and fused_eval(inputs["rgb"], self)
        if overlap and settings.USE_TWO_STREAM_PIPELINE:
            return self._forward_two_streams(inputs, rgb_emb, parts)
        if not overlap:
            _pyr.wait_ready(inputs)
        p_emb, f_pc0 = self._stem_and_mlp1(inputs["cld_rgb_nrm"])             # [B,8,N,1] (+ the first block's mlp1 of it)

        ds_emb = []
        for i_ds in range(4):
            if overlap:
                xyz_i, nei_i, sub_i = inputs["cld_xyz%d" % i_ds], inputs["cld_nei_idx%d" % i_ds], inputs["cld_sub_idx%d" % i_ds]
                with ops.fork(inputs["rgb"].device, 0) as f:
                    if i_ds == 0:
                        _pyr.wait_ready(inputs)                               # the side stream's own wait
                    f.use(p_emb, xyz_i, nei_i, sub_i)
                    f_encoder_i = self.rndla_ds_stages[i_ds](p_emb, xyz_i, nei_i, f_pc=f_pc0 if i_ds == 0 else None)
                    p_emb0 = self.random_sample(f_encoder_i, sub_i)
                rgb_emb0 = self.cnn_ds_stages[i_ds](rgb_emb)
                if i_ds == 0:
                    _pyr.wait_ready(inputs)                                   # the main stream's own wait, before its first index use
                f.join(f_encoder_i, p_emb0)
            else:
                rgb_emb0 = self.cnn_ds_stages[i_ds](rgb_emb)
                f_encoder_i = self.rndla_ds_stages[i_ds](p_emb, inputs["cld_xyz%d" % i_ds], inputs["cld_nei_idx%d" % i_ds],
                                                         f_pc=f_pc0 if i_ds == 0 else None)
                p_emb0 = self.random_sample(f_encoder_i, inputs["cld_sub_idx%d" % i_ds])
            bs, c, hr, wr = rgb_emb0.size()
            if i_ds == 0:
                ds_emb.append(f_encoder_i)

            rgb_emb = self._p2r_fuse(self.ds_fuse_p2r_pre_layers[i_ds], self.ds_fuse_p2r_fuse_layers[i_ds], rgb_emb0, p_emb0,
                                     inputs["p2r_ds_nei_idx%d" % i_ds], want_packed=True)

            r2p_emb = self.random_sample(rgb_emb0.reshape(bs, c, hr * wr), inputs["r2p_ds_nei_idx%d" % i_ds])
            r2p_emb = self.ds_fuse_r2p_pre_layers[i_ds](r2p_emb)
            p_emb = self.ds_fuse_r2p_fuse_layers[i_ds].forward_segs([p_emb0, r2p_emb])       # over cat(p_emb0, r2p_emb), never formed
            ds_emb.append(p_emb)

        n_up = len(self.rndla_up_stages)
        sparse_final = self._sparse_final_ok(inputs["rgb"])
        for i_up in range(n_up - 1):
            rgb_emb0 = self.cnn_up_stages[i_up](rgb_emb)
            bs, c, hr, wr = rgb_emb0.size()

            # decoder layer over cat(skip, nearest_interpolation(p_emb)): the interpolation is the second segment's index
            p_emb0 = self.rndla_up_stages[i_up].forward_segs([ds_emb[-i_up - 2], (p_emb, inputs["cld_interp_idx%d" % (n_up - i_up - 1)])])

            rgb_emb = self._p2r_fuse(self.up_fuse_p2r_pre_layers[i_up], self.up_fuse_p2r_fuse_layers[i_up], rgb_emb0, p_emb0,
                                     inputs["p2r_up_nei_idx%d" % i_up], pixel_major=sparse_final and i_up == n_up - 2)

            r2p_emb = self.random_sample(rgb_emb0.reshape(bs, c, hr * wr), inputs["r2p_up_nei_idx%d" % i_up])
            r2p_emb = self.up_fuse_r2p_pre_layers[i_up](r2p_emb)
            p_emb = self.up_fuse_r2p_fuse_layers[i_up].forward_segs([p_emb0, r2p_emb])

        p_emb = self.rndla_up_stages[n_up - 1].forward_segs([ds_emb[0], (p_emb, inputs["cld_interp_idx0"])]).squeeze(-1)
        if sparse_final:
            # the last stage (up_3 + final) is a per-pixel function of a 3x3 neighbourhood and only the N `choose` pixels of its
            # full-resolution output are kept (reference ffb6d.py:266-285): evaluate it there, on the pixel-major fused map
            rgb_emb_c = self._final_at_choose(rgb_emb, (hr, wr), inputs["choose"])
        elif self._gathered_final_ok():
            # training (and unfused inference): up_3 needs the whole map (its BatchNorm statistics), but FinalStage -- 1x1 convolution +
            # LogSoftmax over channels -- is a per-pixel function, so it commutes with the `choose` gather (reference ffb6d.py:266-285
            # applies it to all H*W pixels and keeps N): gather first, and forward and backward of the stage touch N pixels, not H*W
            last = self.cnn_up_stages[n_up - 1]
            rgb_emb = last[0](rgb_emb)
            bs, di, _, _ = rgb_emb.size()
            rgb_emb_c = ops.gather_nn(rgb_emb.view(bs, di, -1), inputs["choose"].reshape(bs, -1, 1))
            rgb_emb_c = last[1](rgb_emb_c.unsqueeze(-1)).squeeze(-1)
        else:
            rgb_emb = self.cnn_up_stages[n_up - 1](rgb_emb)
            bs, di, _, _ = rgb_emb.size()
            rgb_emb_c = ops.gather_nn(rgb_emb.view(bs, di, -1), inputs["choose"].reshape(bs, -1, 1))
        if parts:
            return rgb_emb_c, p_emb
        return torch.cat([rgb_emb_c, p_emb], dim=1)

    def _stem_and_mlp1(self, x):
        """The RandLA stem fc0 (RandLANet.py:19) and the first block's mlp1 (:683) as ONE launch of two chained per-point layers
        (settings.USE_POINT_CHAIN; same sums in the same order as the two launches) -> (p_emb [B,8,N,1], mlp1(p_emb) [B,16,N,1] or None)."""
        blk = self.rndla_ds_stages[0]
        if settings.USE_POINT_CHAIN and settings.USE_POINTWISE and fused_eval(x, self) and x.dim() == 3:
            p0, p1 = self.rndla_pre_stages._pointwise_params(), blk.mlp1._pointwise_params()
            if (p0 is not None and p1 is not None and p0[0].shape[0] <= 16 and p0[0].shape[1] <= 16 and p1[0].shape[1] <= 32
                    and p1[0].shape[0] == p0[0].shape[1]):
                y0, y1 = ops.pointwise_chain2(x, p0, p1)
                return y0.unsqueeze(3), y1.unsqueeze(3)
        return self.rndla_pre_stages(x).unsqueeze(3), None

    def _forward_two_streams(self, inputs, rgb_emb, parts):
        """The inference forward as a two-stream pipeline (settings.USE_SIDE_STREAMS): the IMAGE stream (the current one) runs the
        trunk / up stages and the point-to-pixel fusions, the POINT stream (side stream 0) the RandLA blocks, the decoder layers and the
        pixel-to-point fusions.  Per stage each stream waits ONCE for the other's product (an event): the point stream for the image
        stage's map (`rgb_emb0`, read by the r2p gather), the image stream for the pooled / decoded point features (`p_emb0`, read by
        the p2r fusion); otherwise they run ahead independently -- the r2p chain of stage i and the RandLA block of stage i + 1 sit in
        the shadow of the convolutions of stage i + 1.  Same kernels on the same operands as the single-stream order: bit-identical
        (tests/test_gpu_headline.py::test_timed_configuration_bit_exact_across_launch_forms).  Every tensor that crosses is recorded
        on the stream that did not allocate it."""
        from . import pyramid as _pyr
        dev = inputs["rgb"].device
        M = torch.cuda.current_stream(dev)
        S = ops.side_stream(dev, 0)

        def to(stream, *ts):
            for t in ts:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(stream)

        def event(stream):
            e = torch.cuda.Event()
            e.record(stream)
            return e

        S.wait_stream(M)
        to(S, inputs["cld_rgb_nrm"])
        with torch.cuda.stream(S):
            _pyr.wait_ready(inputs, cloud_only=True)                          # the point stream's own wait: the cloud's searches only
            p_emb, f_pc0 = self._stem_and_mlp1(inputs["cld_rgb_nrm"])         # [B,8,N,1] (+ the first block's mlp1 of it)
        ds_emb = []
        for i_ds in range(4):
            rgb_emb0 = self.cnn_ds_stages[i_ds](rgb_emb)
            ev_rgb0 = event(M)
            bs, c, hr, wr = rgb_emb0.size()
            with torch.cuda.stream(S):
                f_encoder_i = self.rndla_ds_stages[i_ds](p_emb, inputs["cld_xyz%d" % i_ds], inputs["cld_nei_idx%d" % i_ds],
                                                         f_pc=f_pc0 if i_ds == 0 else None)
                p_emb0 = self.random_sample(f_encoder_i, inputs["cld_sub_idx%d" % i_ds])
                pt = self._p2r_point_term(self.ds_fuse_p2r_pre_layers[i_ds], self.ds_fuse_p2r_fuse_layers[i_ds], c, p_emb0)
                ev_p0 = event(S)
                S.wait_event(ev_rgb0)
                to(S, rgb_emb0)
                if i_ds == 0:
                    _pyr.wait_ready(inputs)                                   # the rest of the pyramid: the r2p / interpolation indices
                r2p_emb = self.random_sample(rgb_emb0.reshape(bs, c, hr * wr), inputs["r2p_ds_nei_idx%d" % i_ds])
                r2p_emb = self.ds_fuse_r2p_pre_layers[i_ds](r2p_emb)
                p_emb = self.ds_fuse_r2p_fuse_layers[i_ds].forward_segs([p_emb0, r2p_emb])
            if i_ds == 0:
                ds_emb.append(f_encoder_i)
            ds_emb.append(p_emb)
            if i_ds == 0:
                _pyr.wait_ready(inputs)                                       # the image stream's own wait: its first index use is this fusion
            M.wait_event(ev_p0)
            to(M, p_emb0, pt)
            rgb_emb = self._p2r_fuse(self.ds_fuse_p2r_pre_layers[i_ds], self.ds_fuse_p2r_fuse_layers[i_ds], rgb_emb0, p_emb0,
                                     inputs["p2r_ds_nei_idx%d" % i_ds], point_term=pt, want_packed=True)
        n_up = len(self.rndla_up_stages)
        sparse_final = self._sparse_final_ok(inputs["rgb"])
        for i_up in range(n_up - 1):
            rgb_emb0 = self.cnn_up_stages[i_up](rgb_emb)
            ev_rgb0 = event(M)
            bs, c, hr, wr = rgb_emb0.size()
            with torch.cuda.stream(S):
                p_emb0 = self.rndla_up_stages[i_up].forward_segs([ds_emb[-i_up - 2], (p_emb, inputs["cld_interp_idx%d" % (n_up - i_up - 1)])])
                pt = self._p2r_point_term(self.up_fuse_p2r_pre_layers[i_up], self.up_fuse_p2r_fuse_layers[i_up], c, p_emb0)
                ev_p0 = event(S)
                S.wait_event(ev_rgb0)
                to(S, rgb_emb0)
                r2p_emb = self.random_sample(rgb_emb0.reshape(bs, c, hr * wr), inputs["r2p_up_nei_idx%d" % i_up])
                r2p_emb = self.up_fuse_r2p_pre_layers[i_up](r2p_emb)
                p_emb = self.up_fuse_r2p_fuse_layers[i_up].forward_segs([p_emb0, r2p_emb])
            M.wait_event(ev_p0)
            to(M, p_emb0, pt)
            rgb_emb = self._p2r_fuse(self.up_fuse_p2r_pre_layers[i_up], self.up_fuse_p2r_fuse_layers[i_up], rgb_emb0, p_emb0,
                                     inputs["p2r_up_nei_idx%d" % i_up], pixel_major=sparse_final and i_up == n_up - 2, point_term=pt)
        with torch.cuda.stream(S):
            p_emb = self.rndla_up_stages[n_up - 1].forward_segs([ds_emb[0], (p_emb, inputs["cld_interp_idx0"])]).squeeze(-1)
        if sparse_final:
            rgb_emb_c = self._final_at_choose(rgb_emb, (hr, wr), inputs["choose"])
        else:
            rgb_emb = self.cnn_up_stages[n_up - 1](rgb_emb)
            bs, di, _, _ = rgb_emb.size()
            rgb_emb_c = ops.gather_nn(rgb_emb.view(bs, di, -1), inputs["choose"].reshape(bs, -1, 1))
        M.wait_stream(S)
        to(M, p_emb)
        if parts:
            return rgb_emb_c, p_emb
        return torch.cat([rgb_emb_c, p_emb], dim=1)

    def _gathered_final_ok(self):
        from .cnn import FinalStage
        last = self.cnn_up_stages[len(self.rndla_up_stages) - 1]
        return settings.USE_GATHERED_FINAL and len(last) == 2 and isinstance(last[1], FinalStage)

    def _sparse_final_ok(self, rgb):
        """Inference with folded BatchNorm, the last stage = PSPUpsample(64 -> 64) + FinalStage(64 -> 64) and the last fusion on the
        64-channel kernel: then the stage runs at the chosen pixels only."""
        from .cnn import FinalStage, PSPUpsample
        last = self.cnn_up_stages[len(self.rndla_up_stages) - 1]
        if not (settings.USE_SPARSE_FINAL and settings.USE_FUSED_UPCONV and fused_eval(rgb, self) and len(last) == 2):
            return False
        up, fin = last[0], last[1]
        if not (isinstance(up, PSPUpsample) and isinstance(fin, FinalStage)):
            return False
        conv, fconv = up.conv[1], fin[0]
        fuse = self.up_fuse_p2r_fuse_layers[len(self.rndla_up_stages) - 2]
        return (conv.in_channels == 64 and conv.out_channels == 64 and fconv.in_channels == 64 and fconv.out_channels == 64
                and act_code(up.conv[3]) is not None and act_code(getattr(fuse, "activation", None)) is not None
                and fuse.conv.weight.shape[0] == 64 and fuse.conv.weight.shape[1] == 128 and rgb.shape[0] <= 65535)

    def _final_at_choose(self, x_pm, hw, choose):
        from .layers import folded_bn as _fbn
        last = self.cnn_up_stages[len(self.rndla_up_stages) - 1]
        up, fin = last[0], last[1]
        conv, fconv = up.conv[1], fin[0]
        key = (conv.weight._version, conv.weight.data_ptr(), fconv.weight._version, fconv.weight.data_ptr())
        cache = self.__dict__.get("_gdm_final_pk")
        if cache is None or cache[0] != key:
            cache = (key, ops.upconv_fused64_pack_weight(conv.weight), ops.pack_rows64(fconv.weight.reshape(64, 64)))
            self.__dict__["_gdm_final_pk"] = cache
        scale, shift = _fbn(up.conv[2], conv.bias)
        code = act_code(up.conv[3])
        return ops.upconv_final_points(x_pm, hw, choose, cache[1], scale, shift, code[0], code[1], cache[2], fconv.bias,
                                       (hw[0] * 2, hw[1] * 2))
