"""64-channel point->pixel fusion tail: hipBLASLt GEMM + gather_add_affine_act vs the one-pass kernel. Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
def tm(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
B, C = 16, 64
for m, n in ((4096, 512), (16384, 128), (16384, 512), (65536, 2048)):
    x = torch.randn(B, C, m, device="cuda"); w = torch.randn(C, C, device="cuda") / 8; t = torch.randn(B, C, n, device="cuda")
    idx = torch.randint(0, n, (B, m), device="cuda", dtype=torch.int32); sc = torch.rand(C, device="cuda") + .5; sh = torch.randn(C, device="cuda")
    wt = w.t().contiguous()
    t_old = tm(lambda: ops.gather_add_affine_act(torch.matmul(w, x), t, idx, sc, sh, 1, 0.0))
    t_new = tm(lambda: ops.conv1x1_gather_add_act(x, wt, t, idx, sc, sh, 1, 0.0))
    print("m=%6d n=%5d: GEMM+tail %7.1f us   one pass %7.1f us   (%.0f MB moved -> %.0f GB/s)" % (m, n, t_old, t_new, 2 * B * C * m * 4 / 1e6, 2 * B * C * m * 4 / t_new / 1e3))
