"""Launches the split-bf16 conv / GEMM kernel a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
x = torch.randn(16, 512, 32, 32, device="cuda"); w = torch.randn(512, 512, 3, 3, device="cuda") / 68.0
wpk = ops.conv3x3_pack_weight(w)
for _ in range(5):
    ops.conv3x3_bf16x3(x, wpk, 512)
x2 = torch.randn(16, 1024, 1024, device="cuda"); w2 = torch.randn(2304, 1024, device="cuda") / 32.0
wpk2 = ops.gemm_pack_weight(w2)
for _ in range(5):
    ops.gemm_bf16x3(x2, wpk2, 2304)
torch.cuda.synchronize()
print("ok")
