"""point_heads_kernel alone at the bench shape (B=16, N=2048).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
B, N = 16, 2048
x0 = torch.randn(B, 128, N, device="cuda")
layers = [(ops.gemm_pack_weight(torch.randn(128, 128, device="cuda") / 11), torch.rand(128, device="cuda") + 0.5, torch.randn(128, device="cuda") * 0.3, 1) for _ in range(8)]
last = (ops.gemm_pack_weight(torch.randn(2, 128, device="cuda") / 11), torch.randn(2, device="cuda"), 2)
f = lambda: ops.point_heads(x0[:, :64].contiguous(), x0[:, 64:].contiguous(), layers, last, 3, 4)
a, b = x0[:, :64].contiguous(), x0[:, 64:].contiguous()
f = lambda: ops.point_heads(a, b, layers, last, 3, 4)
for _ in range(3): f()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20): f()
g.replay(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record(); g.replay(); e.record(); torch.cuda.synchronize()
print("point_heads B=%d N=%d: %.1f us" % (B, N, s.elapsed_time(e) / 20 * 1e3))
