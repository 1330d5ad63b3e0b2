"""A/B of the 3x3 convolution's panel order (GDM_CONV_TAP_INNER=1 chunk-major / 0 tap-major) in ONE process, interleaved rounds, warm chip.
Development aid (round 4)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops, _lib
from geometric_aware_dense_matching_amd._lib import check
L = _lib.lib()
B, H, W = 16, 32, 32


def setup(Cin, Cout):
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5)
    wpk = ops.conv3x3_pack_weight(w)
    xpk = torch.zeros(L.gdm_conv3x3_act_bytes(B, Cin, H, W), dtype=torch.uint8, device="cuda")
    check(L.gdm_conv3x3_pack_act_hip(x.data_ptr(), B, Cin, H, W, xpk.data_ptr(), ops._stream()), "pack")
    out = torch.empty(B, Cout, H, W, device="cuda")
    f = lambda: check(L.gdm_conv3x3_packed_hip(xpk.data_ptr(), wpk.data_ptr(), None, None, None, B, Cin, Cout, H, W, 0, out.data_ptr(), ops._stream()), "conv")
    return f, out, x, w


def tm(f, n=40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for Cin, Cout in ((512, 512), (256, 512), (256, 256), (128, 256), (128, 128)):
    f, out, x, w = setup(Cin, Cout)
    res = {0: [], 1: []}
    outs = {}
    for mode in (0, 1):
        os.environ["GDM_CONV_TAP_INNER"] = str(mode)
        f(); torch.cuda.synchronize()
        outs[mode] = out.clone()
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    err = [float((outs[m].double() - ref).abs().max() / ref.abs().max()) for m in (0, 1)]
    for _ in range(400): f()                       # warm chip
    for r in range(7):
        for mode in (0, 1):
            os.environ["GDM_CONV_TAP_INNER"] = str(mode)
            res[mode].append(tm(f))
    fl = 3 * 2.0 * B * H * W * Cin * Cout * 9
    for mode in (0, 1):
        v = sorted(res[mode])
        print("%4d -> %4d  order %s: median %6.1f us  min %6.1f  (%.2f PF/s issued at the median)  max rel err vs fp64 %.2e" % (
            Cin, Cout, "chunk-major" if mode else "tap-major  ", v[len(v) // 2], v[0], fl / v[len(v) // 2] / 1e9, err[mode]))
