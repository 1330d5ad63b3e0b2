"""Training matching loss at the reference's default training shape (batch 24, N = M = 4096): fused (no similarity matrix) vs the
materialised form, forward + backward, HIP events.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_aware_dense_matching_amd import ops

rs = np.random.RandomState(0)
dev = torch.device("cuda")
B, N, M = 24, 4096, 4096
R = B * N // 2
xyz = torch.from_numpy((rs.rand(M, 3).astype(np.float32) - 0.5) * 0.1).to(dev)
vis = torch.from_numpy((rs.rand(B, M) < 0.6).astype(np.uint8)).to(dev)
g = torch.from_numpy(rs.randint(0, M + 1, size=R).astype(np.int32)).to(dev)
item = torch.from_numpy(np.sort(rs.randint(0, B, size=R)).astype(np.int32)).to(dev)
x0 = torch.nn.functional.normalize(torch.randn(R, 128, device=dev), dim=1)
y0 = torch.nn.functional.normalize(torch.randn(M, 128, device=dev), dim=1)
nbr, visb = ops.circle_nbr_table(xyz, 0.004), ops.circle_visbits(vis)
pad = torch.full((1, 128), -1.0 / np.sqrt(128.0), device=dev)


def fused():
    x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
    ops.circle_match(x, y, g, item, nbr=nbr, visb=visb).sum().backward()


def mat():
    x, y = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
    sim = x @ torch.cat([y, pad], dim=0).t()
    ops.circle_rows(sim, g, item, xyz, vis, 0.004).sum().backward()


for name, fn in (("fused", fused), ("materialised", mat)):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(5):
        fn()
    b.record()
    torch.cuda.synchronize()
    print("%-13s R=%d M=%d: %.3f ms fwd+bwd, peak %.2f GB" % (name, R, M, a.elapsed_time(b) / 5, torch.cuda.max_memory_allocated() / 2**30))
