"""conv(+BN)(+activation) building blocks whose parameter names equal the reference's, so that a
reference checkpoint (`model_state` of /root/reference/train_lm.py:102-117) loads key for key.

Two flavours exist in the reference and both are needed:
  * models/pytorch_utils.py:70-124      submodules `conv`, `normlayer.bn`, `activation`;
    default activation ReLU, torch-default BN (eps 1e-5, momentum 0.1); bias dropped when bn (:90)
  * models/RandLA/pytorch_utils.py:34-99 submodules `conv`, `bn.bn`, `activation`;
    default activation LeakyReLU(0.2); BN eps 1e-6, momentum 0.99 (:103-105)
"""
import torch
import torch.nn as nn

from . import ops, settings


def fused_eval(x, module):
    """True when the inference-only fused BN+activation kernel may replace the module chain: eval mode, GPU
    tensor, autograd off (the kernel has no backward; training and gradient checks use the torch modules)."""
    return (not module.training) and x.is_cuda and not torch.is_grad_enabled()


def folded_bn(bn, conv_bias=None):
    """(scale, shift) of an eval-mode BatchNorm, with an optional preceding conv bias folded in; cached on the
    module and recomputed when any of its tensors changed (version counters / storage)."""
    ts = (bn.weight, bn.bias, bn.running_mean, bn.running_var) + ((conv_bias,) if conv_bias is not None else ())
    key = tuple((t._version, t.data_ptr()) for t in ts)
    cache = bn.__dict__.get("_gdm_fold")
    if cache is None or cache[0] != key:
        with torch.no_grad():
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            shift = bn.bias - bn.running_mean * scale
            if conv_bias is not None:
                shift = shift + conv_bias * scale
        cache = (key, scale.contiguous(), shift.contiguous())
        bn.__dict__["_gdm_fold"] = cache
    return cache[1], cache[2]


def cached_gemm_weight(owner, tag, weight2d, deps):
    """Packed split-bf16 GEMM weight of `weight2d` (a [Cout,Cin] tensor, or a callable producing it), cached on `owner`
    under `tag` until any tensor in `deps` changes."""
    key = tuple((t._version, t.data_ptr()) for t in deps)
    slot = "_gdm_gemm_" + tag
    cache = owner.__dict__.get(slot)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            w = weight2d() if callable(weight2d) else weight2d
            cache = (key, ops.gemm_pack_weight(w.contiguous()), w.shape[0])
        owner.__dict__[slot] = cache
    return cache[1], cache[2]




def act_code(act):
    """(ops.ACT_*, slope) of an activation module; None when it is not one the fused kernel knows."""
    if act is None:
        return ops.ACT_NONE, 0.0
    if isinstance(act, nn.ReLU):
        return ops.ACT_RELU, 0.0
    if isinstance(act, nn.LeakyReLU):
        return ops.ACT_LEAKY, float(act.negative_slope)
    if isinstance(act, nn.PReLU) and act.weight.numel() == 1:
        key = (act.weight._version, act.weight.data_ptr())
        cache = act.__dict__.get("_gdm_slope")
        if cache is None or cache[0] != key:
            cache = (key, float(act.weight.detach().item()))          # one host sync per weight change
            act.__dict__["_gdm_slope"] = cache
        return ops.ACT_LEAKY, cache[1]
    return None


def train_act_code(act):
    """(ops.ACT_*, slope) of an activation the fused training BatchNorm folds in (fixed slopes only); None otherwise."""
    if act is None:
        return ops.ACT_NONE, 0.0
    if isinstance(act, nn.ReLU):
        return ops.ACT_RELU, 0.0
    if isinstance(act, nn.LeakyReLU):
        return ops.ACT_LEAKY, float(act.negative_slope)
    return None


class _FusedConvMixin:
    _bn_name = None

    def _pointwise_params(self):
        """(W^T f32[Cin,Cout], scale, shift, act, slope) of this conv(+BN)(+activation) for ops.pointwise, cached until a parameter
        changes; None when the layer is not a plain 1x1 convolution with an activation the kernel knows."""
        conv = self.conv
        if any(k != 1 for k in conv.kernel_size) or any(st != 1 for st in conv.stride) or conv.groups != 1 or any(p != 0 for p in conv.padding):
            return None
        code = act_code(getattr(self, "activation", None))
        if code is None:
            return None
        bnw = getattr(self, self._bn_name, None)
        deps = [conv.weight] + ([conv.bias] if conv.bias is not None else [])
        if bnw is not None:
            deps += [bnw.bn.weight, bnw.bn.bias, bnw.bn.running_mean, bnw.bn.running_var]
        key = tuple((t._version, t.data_ptr()) for t in deps)
        cache = self.__dict__.get("_gdm_pw")
        if cache is None or cache[0] != key:
            with torch.no_grad():
                wt = conv.weight.reshape(conv.weight.shape[0], -1).t().contiguous()
                if bnw is not None:
                    scale, shift = folded_bn(bnw.bn, conv.bias)
                else:
                    scale, shift = None, (conv.bias.detach().contiguous() if conv.bias is not None else None)
            cache = (key, wt, scale, shift)
            self.__dict__["_gdm_pw"] = cache
        return cache[1], cache[2], cache[3], code[0], code[1]

    def _residual_params(self, other):
        """This layer + `other` (both without activation) as ONE layer over cat(own input, other's input):
        s1*(W1 x1) + b1 + s2*(W2 x2) + b2 = [s1*W1 | s2*W2] . [x1 ; x2] + (b1 + b2).  Cached on self."""
        a, b = self._pointwise_params(), other._pointwise_params()
        if a is None or b is None or a[3] != ops.ACT_NONE or b[3] != ops.ACT_NONE:
            return None
        cache = self.__dict__.get("_gdm_pw_res")
        if cache is None or cache[0] is not a[0] or cache[1] is not b[0] or cache[2] is not a[1] or cache[3] is not b[1]:
            with torch.no_grad():
                def scaled(w, sc):
                    return w if sc is None else w * sc.view(1, -1)
                wt = torch.cat([scaled(a[0], a[1]), scaled(b[0], b[1])], dim=0).contiguous()
                sh = [t for t in (a[2], b[2]) if t is not None]
                shift = (sh[0] + sh[1]) if len(sh) == 2 else (sh[0] if sh else None)
            cache = (a[0], b[0], a[1], b[1], wt, shift)
            self.__dict__["_gdm_pw_res"] = cache
        return cache[4], cache[5]

    def forward_segs(self, segs, res=None, act=None):
        """The layer over cat(segs, dim=1) WITHOUT forming the concat (eval: one ops.pointwise launch).  segs: list of
        x [B,C,n(,1)] or (x [B,C,n_src(,1)], idx [B,n(,1)]) -- an indexed segment is x gathered at idx (nearest interpolation).
        res = (another _FusedConvMixin layer without activation, its input): its output is added before the activation.
        act = (ops.ACT_*, slope) replaces the layer's own activation (the activation after a residual sum).  Returns [B,Cout,n,1]."""
        first = segs[0] if torch.is_tensor(segs[0]) else segs[0][0]
        fused = settings.USE_POINTWISE and fused_eval(first, self)
        pw = self._pointwise_params() if fused else None
        rp = self._residual_params(res[0]) if (pw is not None and res is not None) else None
        if pw is None or (res is not None and rp is None):
            xs = []
            for sp in segs:
                if torch.is_tensor(sp):
                    xs.append(sp if sp.dim() == 4 else sp.unsqueeze(3))
                else:
                    xs.append(ops.gather_nn(sp[0], sp[1]).unsqueeze(3))
            y = self(torch.cat(xs, dim=1) if len(xs) > 1 else xs[0])
            if res is not None:
                rx = res[1]
                y = y + res[0](rx if rx.dim() == 4 else rx.unsqueeze(3))
            if act is not None and act[0] != ops.ACT_NONE:
                y = torch.relu(y) if act[0] == ops.ACT_RELU else torch.nn.functional.leaky_relu(y, negative_slope=act[1])
            return y
        wt, scale, shift, a_code, slope = pw
        if act is not None:
            a_code, slope = act
        segs = list(segs)
        if res is not None:
            wt, shift = rp
            scale = None
            segs.append(res[1])
        return ops.pointwise(segs, wt, scale, shift, a_code, slope).unsqueeze(3)

    def forward(self, x):
        bnw = getattr(self, self._bn_name, None)
        if settings.USE_POINTWISE and fused_eval(x, self) and x.dim() in (3, 4):
            pw = self._pointwise_params()
            if pw is not None:
                B = x.shape[0]
                y = ops.pointwise([x.reshape(B, x.shape[1], -1)], pw[0], pw[1], pw[2], pw[3], pw[4])
                return y.view(B, -1, *x.shape[2:])
        if bnw is not None and fused_eval(x, self):
            code = act_code(getattr(self, "activation", None))
            if code is not None and self.conv.bias is None:
                y = self.conv(x)
                scale, shift = folded_bn(bnw.bn)
                return ops.affine_act(y, scale, shift, code[0], code[1])
        if bnw is not None and self.training:
            act = getattr(self, "activation", None)
            code = train_act_code(act)
            if code is not None:
                y = ops.conv1x1_train(self.conv, x) if ops.conv1x1_train_supported(self.conv, x) else self.conv(x)
                if ops.bn_train_supported(y, bnw.bn):
                    return ops.batch_norm_act_train(y, bnw.bn, code[0], code[1])
                y = bnw(y)
                return act(y) if act is not None else y
        if self.training and ops.conv1x1_train_supported(self.conv, x):
            y = ops.conv1x1_train(self.conv, x)
            for name, mod in self.named_children():
                if name != "conv":
                    y = mod(y)
            return y
        return nn.Sequential.forward(self, x)


class _BN(nn.Sequential):
    def __init__(self, bn_cls, channels, **kw):
        super().__init__()
        self.add_module("bn", bn_cls(channels, **kw))
        nn.init.constant_(self[0].weight, 1.0)
        nn.init.constant_(self[0].bias, 0.0)


class _PtConv(_FusedConvMixin, nn.Sequential):
    """models/pytorch_utils.py:_ConvBase (non-preact form, the only one the hot path uses)."""
    _bn_name = "normlayer"

    def __init__(self, conv_cls, bn_cls, cin, cout, kernel_size, bn, activation, bias):
        super().__init__()
        bias = bias and (not bn)
        conv = conv_cls(cin, cout, kernel_size=kernel_size, bias=bias)
        nn.init.kaiming_normal_(conv.weight)
        if bias:
            nn.init.constant_(conv.bias, 0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("normlayer", _BN(bn_cls, cout))
        if activation is not None:
            self.add_module("activation", activation)


_RELU = object()


def pt_conv1d(cin, cout, bn=False, activation=_RELU, bias=True):
    act = nn.ReLU(inplace=True) if activation is _RELU else activation
    return _PtConv(nn.Conv1d, nn.BatchNorm1d, cin, cout, 1, bn, act, bias)


def pt_conv2d(cin, cout, bn=False, activation=_RELU, bias=True):
    act = nn.ReLU(inplace=True) if activation is _RELU else activation
    return _PtConv(nn.Conv2d, nn.BatchNorm2d, cin, cout, (1, 1), bn, act, bias)


class PtSeq(nn.Sequential):
    """models/pytorch_utils.py:270-316 `Seq(...).conv1d(...)` chains: children named "0", "1", ..."""

    def __init__(self, input_channels):
        super().__init__()
        self.count = 0
        self.current_channels = input_channels

    def conv1d(self, out_size, bn=False, activation=_RELU, bias=True):
        self.add_module(str(self.count), pt_conv1d(self.current_channels, out_size, bn=bn, activation=activation, bias=bias))
        self.count += 1
        self.current_channels = out_size
        return self


class _RlConv(_FusedConvMixin, nn.Sequential):
    """models/RandLA/pytorch_utils.py:_ConvBase (non-preact, no instance norm)."""
    _bn_name = "bn"

    def __init__(self, conv_cls, bn_cls, cin, cout, kernel_size, bn, activation, bias=True):
        super().__init__()
        bias = bias and (not bn)
        conv = conv_cls(cin, cout, kernel_size=kernel_size, bias=bias)
        nn.init.kaiming_normal_(conv.weight)
        if bias:
            nn.init.constant_(conv.bias, 0)
        self.add_module("conv", conv)
        if bn:
            self.add_module("bn", _BN(bn_cls, cout, eps=1e-6, momentum=0.99))
        if activation is not None:
            self.add_module("activation", activation)


_LRELU = object()


def rl_conv1d(cin, cout, bn=False, activation=_LRELU):
    act = nn.LeakyReLU(negative_slope=0.2, inplace=True) if activation is _LRELU else activation
    return _RlConv(nn.Conv1d, nn.BatchNorm1d, cin, cout, 1, bn, act)


def rl_conv2d(cin, cout, bn=False, activation=_LRELU):
    act = nn.LeakyReLU(negative_slope=0.2, inplace=True) if activation is _LRELU else activation
    return _RlConv(nn.Conv2d, nn.BatchNorm2d, cin, cout, (1, 1), bn, act)
