"""Two big K=16 jobs of the pyramid, a few launches each (for rocprofv3 --pmc passes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
torch.manual_seed(0)
for S, Q in ((2048, 2048), (16384, 512)):
    sup = torch.rand(16, S, 3, device="cuda"); qry = torch.rand(16, Q, 3, device="cuda")
    for _ in range(4):
        ops.knn_batch(sup, qry, 16)
torch.cuda.synchronize()
print("ok")
