"""Launches the materialising match kernel a few times (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
B, N, M = 16, 2048, 8192
torch.manual_seed(0)
scene = torch.randn(B, 128, N, device="cuda")
model = torch.randn(128, M, device="cuda")
sim = torch.empty(B, N, M, device="cuda")
srows, mrows = ops.match_pack(scene, 0), ops.match_pack(model, 0)
for _ in range(5):
    ops.match_packed(srows, mrows, B, N, M, 0, return_sim=True, sim_out=sim)
    ops.match_packed(srows, mrows, B, N, M, 0)
torch.cuda.synchronize()
print("ok")
