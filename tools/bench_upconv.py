import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
for Cout, H in ((256, 32), (64, 64), (64, 128)):
    z = torch.randn(16, 9 * Cout, H, H, device="cuda"); sc = torch.ones(Cout, device="cuda"); sh = torch.zeros(Cout, device="cuda")
    for _ in range(3): ops.upconv3x3_gather(z, sc, sh, Cout, (2 * H, 2 * H), ops.ACT_LEAKY, 0.25)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(10): ops.upconv3x3_gather(z, sc, sh, Cout, (2 * H, 2 * H), ops.ACT_LEAKY, 0.25)
    b.record(); torch.cuda.synchronize()
    print("upconv gather Cout=%3d %3d->%3d: %.1f us" % (Cout, H, 2 * H, a.elapsed_time(b) * 100))
