"""conv3x3 split-bf16 MFMA kernel vs MIOpen fp32 on the ResNet layer3/4 shapes (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_aware_dense_matching_amd import ops
torch.backends.cudnn.benchmark = True
B = 16
def tm(f, n=10):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for Cin, Cout in ((128, 256), (256, 256), (256, 512), (512, 512)):
    x = torch.randn(B, Cin, 32, 32, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    wpk = ops.conv3x3_pack_weight(w)
    t_mi = tm(lambda: torch.nn.functional.conv2d(x, w, padding=1))
    t_us = tm(lambda: ops.conv3x3_bf16x3(x, wpk, Cout))
    gf = 2.0 * B * 1024 * Cin * Cout * 9 / 1e9
    print("%4d->%4d: MIOpen %7.1f us (%6.1f TF/s)   bf16x3 %7.1f us (%6.1f TF/s fp32-equivalent)" % (Cin, Cout, t_mi, gf / t_mi * 1e-3 * 1e3 / 1e3 * 1e3, t_us, gf / t_us))
print("--- 1x1 / GEMM ---")
for Cin, Cout, n, Bb in ((1024, 2304, 1024, 16), (512, 1024, 1024, 16), (1024, 1024, 1024, 16), (512, 512, 1024, 16), (128, 16000, 8192, 1)):
    x = torch.randn(Bb, Cin, n, device="cuda"); w = torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5
    wpk = ops.gemm_pack_weight(w)
    t_mi = tm(lambda: torch.matmul(w, x))
    t_us = tm(lambda: ops.gemm_bf16x3(x, wpk, Cout))
    gf = 2.0 * Bb * n * Cin * Cout / 1e9
    print("%4d->%5d n=%5d B=%2d: hipBLASLt %7.1f us   bf16x3 %7.1f us (%6.1f TF/s fp32-equivalent)" % (Cin, Cout, n, Bb, t_mi, t_us, gf / t_us * 1e3 / 1e3))
