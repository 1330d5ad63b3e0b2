"""ops.wgrad_direct (one-pass small-channel weight + bias gradient) vs the batched fp32 GEMM + sums it replaces.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops

def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

for (B, Cin, Cout, P) in [(24, 32, 32, 65536), (24, 64, 64, 16384), (24, 128, 64, 16384), (24, 64, 128, 16384), (24, 128, 128, 4096), (24, 64, 64, 4096),
                          (24, 32, 64, 4096), (24, 16, 32, 65536), (24, 8, 16, 65536), (24, 64, 32, 65536), (24, 256, 256, 1024), (24, 256, 128, 4096)]:
    x = torch.randn(B, Cin, P, device="cuda"); go = torch.randn(B, Cout, P, device="cuda")
    own = t(lambda: ops.wgrad_direct(x, go, bias=True))
    lib = t(lambda: (torch.bmm(go, x.transpose(1, 2)).sum(0), go.sum((0, 2))))
    libw = t(lambda: torch.bmm(go, x.transpose(1, 2)).sum(0))
    mb = (x.numel() + go.numel()) * 4 / 1e6
    print("B=%d Cin=%d Cout=%d P=%d: own %.1f us (%.2f TB/s)  bmm+sum+bias %.1f us  bmm+sum %.1f us" % (B, Cin, Cout, P, own, mb / own, lib, libw))
