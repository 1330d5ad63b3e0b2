"""Two shapes of the split-bf16 conv / GEMM kernel, kernel only (activations packed once). Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops, _lib
from geometric_aware_dense_matching_amd._lib import check
def tm(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
L = _lib.lib()
B, Cin, Cout, H, W = 16, 512, 512, 32, 32
x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / 68
wpk = ops.conv3x3_pack_weight(w)
xpk = torch.zeros(L.gdm_conv3x3_act_bytes(B, Cin, H, W), dtype=torch.uint8, device="cuda")
check(L.gdm_conv3x3_pack_act_hip(x.data_ptr(), B, Cin, H, W, xpk.data_ptr(), ops._stream()), "pack")
out = torch.empty(B, Cout, H, W, device="cuda")
t = tm(lambda: check(L.gdm_conv3x3_packed_hip(xpk.data_ptr(), wpk.data_ptr(), None, None, None, B, Cin, Cout, H, W, 0, out.data_ptr(), ops._stream()), "conv"))
fl = 2.0 * B * H * W * Cin * Cout * 9
print("conv3x3 512->512 B16 32x32 : %7.1f us  %6.1f TF/s fp32-equivalent, %6.1f TF/s bf16 issued (%.0f%% of 2.5 PF)" % (t, fl / t / 1e6, 3 * fl / t / 1e6, 3 * fl / t / 1e6 / 25))
Cin, Cout = 1024, 2304
x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, device="cuda") / 32
wpk = ops.gemm_pack_weight(w)
xpk = torch.zeros(L.gdm_conv3x3_act_bytes(B, Cin, H, W), dtype=torch.uint8, device="cuda")
check(L.gdm_conv3x3_pack_act_hip(x.data_ptr(), B, Cin, H, W, xpk.data_ptr(), ops._stream()), "pack")
out = torch.empty(B, Cout, H, W, device="cuda")
t = tm(lambda: check(L.gdm_conv1x1_packed_hip(xpk.data_ptr(), wpk.data_ptr(), None, None, B, Cin, Cout, H, W, 0, 0, out.data_ptr(), ops._stream()), "gemm"))
fl = 2.0 * B * H * W * Cin * Cout
print("gemm 1024->2304 n=16x1024  : %7.1f us  %6.1f TF/s fp32-equivalent, %6.1f TF/s bf16 issued (%.0f%% of 2.5 PF)" % (t, fl / t / 1e6, 3 * fl / t / 1e6, 3 * fl / t / 1e6 / 25))
