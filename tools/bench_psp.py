import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
g = torch.randn(16, 512, 32, 32, device="cuda")
ys = [torch.randn(16, 512, s, s, device="cuda") for s in (1, 2, 3, 6)]
bias = torch.randn(512, device="cuda")
def tm(f, n=20):
    for _ in range(3): f()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); a.record()
    for _ in range(n): f()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print("psp_combine %.1f us" % tm(lambda: ops.psp_combine(g, ys, bias)))
