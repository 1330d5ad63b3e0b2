"""Fused attentive-pooling stages vs the separate-kernel chain on the four RandLA levels. Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import randla, settings
def tm(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, K = 16, 16
for d_out, n in ((32, 2048), (64, 512), (128, 128), (256, 32)):
    blk = randla.BuildingBlock(d_out).cuda().eval()
    xyz = torch.randn(B, n, 3, device="cuda"); feat = torch.randn(B, d_out // 2, n, 1, device="cuda")
    idx = torch.randint(0, n, (B, n, K), device="cuda", dtype=torch.int32)
    with torch.no_grad():
        settings.USE_FUSED_LFA = False; t0 = tm(lambda: blk(xyz, feat, idx))
        settings.USE_FUSED_LFA = True; t1 = tm(lambda: blk(xyz, feat, idx))
    print("d_out=%3d n=%4d: separate kernels %7.1f us   fused (2 launches) %7.1f us" % (d_out, n, t0, t1))
