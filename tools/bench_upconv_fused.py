"""PSPUpsample(64,64) at the last up stage: two-kernel form (low-resolution GEMM + gather) vs the one-kernel form. Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import cnn, settings
def tm(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
mod = cnn.PSPUpsample(64, 64).cuda().eval()
for B, H in ((16, 128), (16, 64), (1, 128)):
    x = torch.randn(B, 64, H, H, device="cuda")
    with torch.no_grad():
        settings.USE_FUSED_UPCONV = False; t0 = tm(lambda: mod(x))
        settings.USE_FUSED_UPCONV = True; t1 = tm(lambda: mod(x))
    print("B=%2d %3d->%3d: two kernels %7.1f us   fused %7.1f us" % (B, H, 2 * H, t0, t1))
