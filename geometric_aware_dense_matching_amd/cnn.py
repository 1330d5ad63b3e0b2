"""Image branch: ResNet-18 trunk + pyramid pooling + three x2 upsample stages.

Stays on PyTorch-ROCm (MIOpen convolutions) -- SURVEY.md row a7: not a custom-kernel row.
Module / parameter names equal the reference's so checkpoints load key for key:
  /root/reference/models/cnn/extractors.py:107-200  ResNet(BasicBlock, [2,2,2,2])
  /root/reference/models/cnn/pspnet.py:7-45,93-138  PSPModule, PSPUpsample, PSPNet

Behaviour worth knowing (all reproduced):
  * `_make_layer` is called with dilation=2/4 for layer3/4 but passes `self.current_dilation`
    (still 1, because output_stride defaults to 32) to the blocks, so the trunk is a plain
    stride-8 ResNet with NO dilation (extractors.py:151-177).
  * `final` = Conv2d(64,64,1) + nn.LogSoftmax() with implicit dim, which is dim=1 for 4-D input
    (pspnet.py:108-112).
  * No ImageNet download: weights come from the checkpoint (the reference fetches them over the
    network at construction, pspnet.py:141 -> extractors.py:203-212).
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, settings
from .layers import act_code, cached_gemm_weight, folded_bn, fused_eval




def bn_act(bn, x, relu=None):
    """relu(bn(x)) (relu optional): the fused training kernels where they apply, the modules otherwise."""
    if ops.bn_train_supported(x, bn):
        return ops.batch_norm_act_train(x, bn, ops.ACT_RELU if relu is not None else ops.ACT_NONE)
    y = bn(x)
    return relu(y) if relu is not None else y


def _conv3x3(cin, cout, stride=1, dilation=1):
    return nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=dilation, dilation=dilation, bias=False)


class BasicBlock(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = _conv3x3(inplanes, planes, stride=stride, dilation=dilation)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _conv3x3(planes, planes, stride=1, dilation=dilation)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    @staticmethod
    def _packed_weight(conv):
        w = conv.weight
        key = (w._version, w.data_ptr())
        cache = conv.__dict__.get("_gdm_wpk")
        if cache is None or cache[0] != key:
            cache = (key, ops.conv3x3_pack_weight(w))
            conv.__dict__["_gdm_wpk"] = cache
        return cache[1]

    @staticmethod
    def _train_conv(conv, x):
        """Training: forward and input gradient of the 32x32-resolution 3x3 convolutions on the split-bf16 MFMA kernel (MIOpen's fp32
        Winograd runs them 3x slower); the weight gradient stays with MIOpen."""
        if (settings.USE_MFMA_CONV_TRAIN and conv.bias is None and tuple(conv.stride) == (1, 1)
                and ops.conv3x3_supported(x, conv.weight, conv.stride, conv.padding, conv.dilation)):
            return ops.conv3x3_train(x, conv.weight)
        return conv(x)

    def _mfma_ok(self, x):
        if tuple(self.conv1.stride) != (1, 1) and not settings.USE_MFMA_STRIDED:
            return False
        return (settings.USE_MFMA_CONV and ops.conv3x3_supported(x, self.conv1.weight, self.conv1.stride, self.conv1.padding, self.conv1.dilation)
                and ops.conv3x3_supported(x, self.conv2.weight, self.conv2.stride, self.conv2.padding, self.conv2.dilation)
                and self.conv2.weight.shape[1] == self.conv1.weight.shape[0])

    def forward(self, x):
        if fused_eval(x, self) and self._mfma_ok(x):
            # 64 x 64 and 32 x 32 stages: both 3x3 convolutions on the split-bf16 MFMA implicit-GEMM kernel with BN, ReLU and the
            # residual add in its epilogue (MIOpen's fp32 path here is a vector-ALU Winograd kernel, 2.7x slower).  The epilogue
            # also writes the NEXT convolution's operand (bf16 hi / lo planes): conv1 -> conv2 needs no fp32 map at all, and a block
            # hands its packed output to the next block on the tensor it returns (`_gdm_packed`), so one pack launch feeds a whole
            # chain of blocks.
            s1, b1 = folded_bn(self.bn1)
            planes = self.conv1.weight.shape[0]
            stride = self.conv1.stride[0]
            xin = getattr(x, "_gdm_packed", None)
            if xin is None or xin.shape != tuple(x.shape):
                xin = x
            dconv = self.downsample[0] if self.downsample is not None else None
            own_ds = (settings.USE_MFMA_STRIDED and dconv is not None and isinstance(dconv, nn.Conv2d) and dconv.kernel_size == (1, 1) and dconv.bias is None
                      and dconv.stride == self.conv1.stride and dconv.padding == (0, 0)
                      and ops.gemm_supported(dconv.in_channels, dconv.out_channels, 64))
            if own_ds and not isinstance(xin, ops.PackedAct):
                xin = ops.conv3x3_pack_act(x)                      # one packed operand feeds conv1 and the downsample branch
            chain = (x.shape[0] * (x.shape[2] // stride) * (x.shape[3] // stride)) % 256 == 0
            mid = ops.conv3x3_bf16x3(xin, self._packed_weight(self.conv1), planes, s1, b1, ops.ACT_RELU, out_f32=not chain, out_packed=chain,
                                     stride=stride)
            mid = mid[1] if chain else mid
            s2, b2 = folded_bn(self.bn2)
            if self.downsample is None:
                res = x
            else:
                sd, bd = folded_bn(self.downsample[1])
                if own_ds:
                    # 1x1 (strided) convolution + BN on the same kernel, reading the packed map conv1 reads
                    wd, _ = cached_gemm_weight(dconv, "ds", lambda: dconv.weight.reshape(dconv.out_channels, dconv.in_channels), (dconv.weight,))
                    res = ops.conv1x1_packed2d(xin, wd, dconv.out_channels, sd, bd, ops.ACT_NONE, stride=stride)
                else:
                    res = ops.affine_act(dconv(x), sd, bd, ops.ACT_NONE)
            if not chain:
                return ops.conv3x3_bf16x3(mid, self._packed_weight(self.conv2), planes, s2, b2, ops.ACT_RELU, res)
            out, opk = ops.conv3x3_bf16x3(mid, self._packed_weight(self.conv2), planes, s2, b2, ops.ACT_RELU, res, out_packed=True)
            out._gdm_packed = opk
            return out
        if fused_eval(x, self):
            # eval: conv -> [BN+ReLU] -> conv -> [BN + (BN'd) residual + ReLU], each bracket one HIP launch
            s1, b1 = folded_bn(self.bn1)
            out = ops.affine_act(self.conv1(x), s1, b1, ops.ACT_RELU)
            s2, b2 = folded_bn(self.bn2)
            if self.downsample is None:
                return ops.affine_act(self.conv2(out), s2, b2, ops.ACT_RELU, res=x)
            sd, bd = folded_bn(self.downsample[1])
            return ops.affine_act(self.conv2(out), s2, b2, ops.ACT_RELU, res=self.downsample[0](x), res_scale=sd, res_shift=bd)
        out = bn_act(self.bn1, self._train_conv(self.conv1, x), self.relu)
        out = bn_act(self.bn2, self._train_conv(self.conv2, out))
        residual = x if self.downsample is None else bn_act(self.downsample[1], self.downsample[0](x))
        out = out + residual
        return self.relu(out)


class ResNet18Trunk(nn.Module):
    def __init__(self):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(64, 2, stride=1)
        self.layer2 = self._make_layer(128, 2, stride=2)
        self.layer3 = self._make_layer(256, 2, stride=1)     # reference asks for dilation 2, gets 1
        self.layer4 = self._make_layer(512, 2, stride=1)     # reference asks for dilation 4, gets 1
        self.fc = nn.Linear(512, 1000)                       # unused; kept for checkpoint key parity
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, planes, blocks, stride):
        downsample = None
        if stride != 1 or self.inplanes != planes:
            downsample = nn.Sequential(nn.Conv2d(self.inplanes, planes, kernel_size=1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes))
        layers = [BasicBlock(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes
        for _ in range(1, blocks):
            layers.append(BasicBlock(self.inplanes, planes))
        return nn.Sequential(*layers)


class PSPModule(nn.Module):
    def __init__(self, features, out_features=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(output_size=(s, s)),
                                                   nn.Conv2d(features, features, kernel_size=1, bias=False))
                                     for s in sizes])
        self.bottleneck = nn.Conv2d(features * (len(sizes) + 1), out_features, kernel_size=1)
        self.relu = nn.ReLU()

    def _split_weights(self):
        """Bottleneck weight split over the concat [prior_1..prior_4, feats] with each prior's own 1x1 conv folded in:
        M_k = W[:, kF:(k+1)F] @ V_k (so W_k . up(V_k pool_k f) = up(M_k pool_k f)), W_f = W[:, 4F:].  Cached."""
        w = self.bottleneck.weight
        vs = [st[1].weight for st in self.stages]
        key = tuple((t._version, t.data_ptr()) for t in [w] + vs)
        cache = self.__dict__.get("_gdm_split")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                F_ = vs[0].shape[0]
                w2 = w.view(w.shape[0], -1)
                ms = [(w2[:, k * F_:(k + 1) * F_] @ v.view(F_, F_)).contiguous() for k, v in enumerate(vs)]
                wf = w2[:, len(vs) * F_:].contiguous()
            cache = (key, ms, wf)
            self.__dict__["_gdm_split"] = cache
        return cache[1], cache[2]

    def _split_weights_t(self, ms):
        """The folded prior matrices transposed ([Cin, Cout], the per-point kernel's weight layout), cached beside them."""
        cache = self.__dict__.get("_gdm_split_t")
        if cache is None or cache[0] is not ms:
            with torch.no_grad():
                cache = (ms, [m.t().contiguous() for m in ms])
            self.__dict__["_gdm_split_t"] = cache
        return cache[1]

    def forward(self, feats):
        h, w = feats.size(2), feats.size(3)
        if fused_eval(feats, self) and len(self.stages) == 4 and feats.shape[0] * self.bottleneck.out_channels <= 65535:
            # relu(W cat(up(p_k)..., f) + b) = relu(W_f f + b + sum_k up(M_k pool_k f)): K = 512 instead of 2560, no concat,
            # no full-resolution priors
            ms, wf = self._split_weights()
            B, Cin = feats.shape[0], feats.shape[1]
            def full_res():
                if settings.USE_MFMA_GEMM and ops.gemm_supported(Cin, wf.shape[0], h * w):
                    wpk, co = cached_gemm_weight(self, "wf", wf, (self.bottleneck.weight,))
                    return ops.gemm_bf16x3_map(feats, wpk, co)               # reads layer4's packed output when it is there
                return ops.wx(wf, feats.reshape(B, Cin, h * w)).view(B, -1, h, w)

            def priors():
                sizes = [st[0].output_size[0] if isinstance(st[0].output_size, (tuple, list)) else st[0].output_size for st in self.stages]
                pools = ops.psp_pools(feats) if sizes == [1, 2, 3, 6] and ops.psp_pools_supported(h, w) else None
                ys = []
                mts = self._split_weights_t(ms) if settings.USE_POINTWISE else None
                if mts is not None and pools is not None and settings.USE_PSP_JOBS and Cin >= 32:
                    # the four prior products M_k . pool_k(f) in one launch (each alone is a latency-bound launch on 16 .. 576 points)
                    outs = ops.pointwise_jobs([p.reshape(B, Cin, -1) for p in pools], mts)
                    return [o.view(B, -1, p.shape[2], p.shape[3]) for o, p in zip(outs, pools)]
                for k, (st, m) in enumerate(zip(self.stages, ms)):
                    p = pools[k] if pools is not None else st[0](feats)        # adaptive average pool to s x s
                    s_ = p.shape[2]
                    if mts is not None:
                        ys.append(ops.pointwise([p.reshape(B, Cin, s_ * s_)], mts[k]).view(B, -1, s_, s_))   # M_k . pool_k(f): own kernel
                    else:
                        ys.append(ops.wx(m, p.reshape(B, Cin, s_ * s_)).view(B, -1, s_, s_))
                return ys

            if settings.USE_SIDE_STREAMS and "psp" in settings.SIDE_PARTS and not torch.is_grad_enabled():
                # the pools and the four prior products (five launches on a few hundred points) beside the full-resolution GEMM: both read
                # the same map and meet in psp_combine -- same kernels on the same operands as the sequential order
                with ops.fork(feats.device, 3) as f:           # 0 = point branch, 1 = mesh branch, 2 = pyramid
                    f.use(feats)
                    ys = priors()
                g = full_res()
                f.join(*ys)
            else:
                g = full_res()
                ys = priors()
            return ops.psp_combine(g, ys, self.bottleneck.bias, packed=settings.USE_MFMA_GEMM and settings.USE_PACKED_PRODUCERS)
        sizes = [st[0].output_size[0] if isinstance(st[0].output_size, (tuple, list)) else st[0].output_size for st in self.stages]
        if (settings.USE_SPLIT_PSP_TRAIN and feats.is_cuda and feats.dtype == torch.float32 and sizes == [1, 2, 3, 6] and ops.psp_pools_supported(h, w)
                and (h * w) % 4 == 0 and feats.shape[0] * self.bottleneck.out_channels <= 65535 and h <= 64 and w <= 64):
            # training: the same algebra as the eval path, under autograd.  W cat(up(V_k pool_k f)..., f) = W_f f + sum_k up((W_k V_k) pool_k f):
            # the big GEMM has K = F instead of 5F in forward, dgrad and wgrad, the four full-resolution priors and their concat are
            # never built, one launch pools, one launch adds the priors + bias + ReLU; their backwards are single-pass kernels too
            B, F_ = feats.shape[0], feats.shape[1]
            w2 = self.bottleneck.weight.view(self.bottleneck.out_channels, -1)
            g = ops.wx(w2[:, 4 * F_:], feats.reshape(B, F_, h * w)).view(B, -1, h, w)
            pools = ops.psp_pools(feats)
            ys = []
            for k, st in enumerate(self.stages):
                m = w2[:, k * F_:(k + 1) * F_] @ st[1].weight.view(F_, F_)
                ys.append(ops.wx(m, pools[k].reshape(B, F_, sizes[k] * sizes[k])).view(B, -1, sizes[k], sizes[k]))
            return ops.psp_combine_train(g, ys, self.bottleneck.bias)
        priors = [ops.upsample_bilinear(stage(feats), (h, w)) for stage in self.stages] + [feats]
        return self.relu(self.bottleneck(torch.cat(priors, 1)))


class Upsample2x(nn.Module):
    """nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True) on the HIP kernel (no parameters,
    so it keeps slot 0 of PSPUpsample.conv and the checkpoint keys conv.1/2/3 unchanged)."""

    def forward(self, x):
        return ops.upsample_bilinear(x, (x.shape[2] * 2, x.shape[3] * 2))


class PSPUpsample(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(Upsample2x(), nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.PReLU())

    def _tap_major_weight(self):
        """[Cout,Cin,3,3] -> [9*Cout,Cin], row = tap*Cout + co; cached until the weight changes."""
        w = self.conv[1].weight
        key = (w._version, w.data_ptr())
        cache = self.__dict__.get("_gdm_wt")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                wt = w.permute(2, 3, 0, 1).reshape(9 * w.shape[0], w.shape[1]).contiguous()
            cache = (key, wt)
            self.__dict__["_gdm_wt"] = cache
        return cache[1]

    def forward(self, x):
        if fused_eval(x, self):
            code = act_code(self.conv[3])
            conv = self.conv[1]
            if code is not None and x.shape[0] * conv.out_channels <= 65535 and conv.in_channels >= settings.UPCONV_MIN_CIN:
                # conv3x3(up(x)) = 9-tap bilinear gather of a LOW-resolution 1x1 convolution (4x fewer FLOPs, no
                # 2x-resolution intermediate), BN (with the conv bias) + PReLU folded into the gather's epilogue
                Bx, Cin, Hx, Wx = x.shape
                if settings.USE_FUSED_UPCONV and Cin == 64 and conv.out_channels == 64 and Hx >= 2 and Wx >= 2 and Bx <= 65535:
                    # 64 -> 64 (last up stage): channel mix on the matrix cores into LDS + gather in ONE kernel, no 9*64-channel tensor
                    w = conv.weight
                    key = (w._version, w.data_ptr())
                    cache = self.__dict__.get("_gdm_fused64")
                    if cache is None or cache[0] != key:
                        cache = (key, ops.upconv_fused64_pack_weight(w))
                        self.__dict__["_gdm_fused64"] = cache
                    scale, shift = folded_bn(self.conv[2], conv.bias)
                    return ops.upconv_fused64(x, cache[1], scale, shift, (Hx * 2, Wx * 2), code[0], code[1])
                if settings.USE_MFMA_GEMM and ops.gemm_supported(Cin, 9 * conv.out_channels, Hx * Wx):
                    wpk, c9 = cached_gemm_weight(self, "tap", self._tap_major_weight, (conv.weight,))
                    z = ops.gemm_bf16x3_map(x, wpk, c9)            # split-bf16 MFMA; reads the packed operand its producer wrote, if any
                else:
                    z = ops.wx(self._tap_major_weight(), x.reshape(Bx, Cin, Hx * Wx)).view(Bx, -1, Hx, Wx)   # hipBLASLt GEMM
                scale, shift = folded_bn(self.conv[2], conv.bias)
                # >= 128 channels: the next reader is a GEMM over the map (the p2r fusion's pixel half), so the gather writes its operand
                return ops.upconv3x3_gather(z, scale, shift, conv.out_channels, (x.shape[2] * 2, x.shape[3] * 2), code[0], code[1],
                                            packed=settings.USE_MFMA_GEMM and settings.USE_PACKED_PRODUCERS and conv.out_channels % 128 == 0)
        act = self.conv[3]
        if x.is_cuda and isinstance(act, nn.PReLU) and act.weight.numel() == 1 and x.dtype == torch.float32:
            conv = self.conv[1]
            if settings.USE_LOWRES_UPCONV_TRAIN and ops.upconv_train_supported(x.shape[0], conv.out_channels) and conv.in_channels >= settings.UPCONV_MIN_CIN:
                # training too: the 3x3 convolution on the upsampled map as a low-resolution GEMM (autograd) + the differentiable
                # 9-tap gather: 4x fewer FLOPs forward and backward, no 2x-resolution input, MIOpen's fp32 wrw / bwd-data not needed
                Bx, Cin, Hx, Wx = x.shape
                w9 = conv.weight.permute(2, 3, 0, 1).reshape(9 * conv.out_channels, Cin)
                z = ops.wx(w9, x.reshape(Bx, Cin, Hx * Wx)).view(Bx, -1, Hx, Wx)
                y = bn_act(self.conv[2], ops.upconv3x3_gather_train(z, conv.bias, conv.out_channels, (Hx * 2, Wx * 2)))
            else:
                y = bn_act(self.conv[2], conv(self.conv[0](x)))
            if y.numel() % 4 == 0:
                return ops.prelu1(y, act.weight)           # torch's PReLU backward runs at ~0.6 TB/s on these 10^8-element maps
            return act(y)
        return self.conv(x)


class FinalStage(nn.Sequential):
    """Conv2d(64,64,1) + LogSoftmax(dim=1) (pspnet.py:108-112; same child names "0", "1"): one HIP pass in eval."""

    def forward(self, x):
        conv = self[0]
        if fused_eval(x, self) and conv.in_channels == 64 and conv.out_channels == 64 and x.shape[0] <= 65535:
            return ops.conv1x1_logsoftmax(x, conv.weight, conv.bias)
        if self.training and ops.conv1x1_train_supported(conv, x):
            return self[1](ops.conv1x1_train(conv, x))
        return nn.Sequential.forward(self, x)


class PSPNet(nn.Module):
    """pspnet.py:93-121 with the resnet18 settings of `psp_models['resnet18']` (:141); only the
    sub-modules FFB6DEmb borrows are used, the rest exists for checkpoint key parity."""

    def __init__(self):
        super().__init__()
        self.feats = ResNet18Trunk()
        self.psp = PSPModule(512, 1024, (1, 2, 3, 6))
        self.drop_1 = nn.Dropout2d(p=0.3)
        self.up_1 = PSPUpsample(1024, 256)
        self.up_2 = PSPUpsample(256, 64)
        self.up_3 = PSPUpsample(64, 64)
        self.drop_2 = nn.Dropout2d(p=0.15)
        self.final = FinalStage(nn.Conv2d(64, 64, kernel_size=1), nn.LogSoftmax(dim=1))
