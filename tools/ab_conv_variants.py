"""A/B of compile-time variants of the convolution / GEMM kernel in ONE process: every variant is its own build of the library
(tools/micro/variants/libgdm_<name>.so: gdm_conv.hip compiled with extra -D flags and linked with the shipped objects), loaded side by
side with ctypes; interleaved rounds on a warm chip, median and minimum per variant.  Development aid (round 4).
    python tools/ab_conv_variants.py [name ...]        (default: the shipped library + every variant found)"""
import ctypes, glob, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from geometric_aware_dense_matching_amd import ops, _lib
from geometric_aware_dense_matching_amd._lib import check
L0 = _lib.lib()
libs = {"shipped": L0}
for path in sorted(glob.glob(os.path.join(ROOT, "tools", "micro", "variants", "libgdm_*.so"))):
    name = os.path.basename(path)[len("libgdm_"):-3]
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    l = ctypes.CDLL(path)
    for fn in ("gdm_conv3x3_packed_hip", "gdm_conv1x1_packed_hip"):
        getattr(l, fn).restype = _lib.SIGNATURES[fn][0]
        getattr(l, fn).argtypes = _lib.SIGNATURES[fn][1]
    libs[name] = l
B, H, W = 16, 32, 32


def tm(f, n=40):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for taps, Cin, Cout in ((9, 512, 512), (9, 256, 256), (9, 128, 128), (1, 1024, 2304), (1, 512, 1024)):
    x = torch.randn(B, Cin, H, W, device="cuda")
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (3 * Cin ** 0.5) if taps == 9 else torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5
    wpk = ops.conv3x3_pack_weight(w) if taps == 9 else ops.gemm_pack_weight(w)
    xpk = torch.zeros(L0.gdm_conv3x3_act_bytes(B, Cin, H, W), dtype=torch.uint8, device="cuda")
    check(L0.gdm_conv3x3_pack_act_hip(x.data_ptr(), B, Cin, H, W, xpk.data_ptr(), ops._stream()), "pack")
    out = torch.empty(B, Cout, H, W, device="cuda")
    fs, outs = {}, {}
    for name, l in libs.items():
        if taps == 9:
            fs[name] = (lambda l=l: l.gdm_conv3x3_packed_hip(xpk.data_ptr(), wpk.data_ptr(), None, None, None, B, Cin, Cout, H, W, 0, out.data_ptr(), ops._stream()))
        else:
            fs[name] = (lambda l=l: l.gdm_conv1x1_packed_hip(xpk.data_ptr(), wpk.data_ptr(), None, None, B, Cin, Cout, H, W, 0, 0, out.data_ptr(), ops._stream()))
        assert fs[name]() == 0
        torch.cuda.synchronize()
        outs[name] = out.clone()
    same = {n: bool(torch.equal(outs[n], outs["shipped"])) for n in libs}
    for _ in range(300): fs["shipped"]()
    res = {n: [] for n in libs}
    for r in range(7):
        for n in libs:
            res[n].append(tm(fs[n]))
    fl = 3 * 2.0 * B * H * W * Cin * Cout * taps
    for n in libs:
        v = sorted(res[n])
        print("%s %4d -> %4d  %-8s median %6.1f us  min %6.1f  (%.2f PF/s issued)  == shipped: %s" % (
            "3x3" if taps == 9 else "1x1", Cin, Cout, n, v[len(v) // 2], v[0], fl / v[len(v) // 2] / 1e9, same[n]))
