"""Per-job timing of the HIP kNN (development aid): python tools/bench_knn.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops

B = 16
jobs = [(2048, 2048, 16), (512, 2048, 1), (4096, 512, 16), (512, 4096, 1), (512, 512, 16), (1024, 128, 16),
        (128, 128, 16), (1024, 32, 16), (32, 32, 16), (1024, 8, 16), (4096, 32, 16), (16384, 128, 16), (128, 16384, 1),
        (16384, 512, 16), (512, 16384, 1)]
torch.manual_seed(0)
for S, Q, K in jobs:
    sup = torch.rand(B, S, 3, device="cuda")
    qry = torch.rand(B, Q, 3, device="cuda")
    for _ in range(3):
        ops.knn_batch(sup, qry, K)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.knn_batch(sup, qry, K)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print("S=%6d Q=%6d K=%2d  %8.1f us  %7.1f Gpairs/s" % (S, Q, K, ms * 1e3, B * S * Q / ms / 1e6))
