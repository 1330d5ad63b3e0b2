"""Per-shape timing of the K=16 / K=1 searches of the neighbour pyramid (B crops each), one launch per shape, HIP events around
20 back-to-back launches.  Development aid: python tools/bench_knn_jobs.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
torch.manual_seed(0)
shapes = [(2048, 2048, 16), (4096, 512, 16), (512, 512, 16), (1024, 128, 16), (128, 128, 16), (1024, 32, 16), (32, 32, 16),
          (1024, 8, 16), (4096, 32, 16), (16384, 128, 16), (16384, 512, 16), (512, 16384, 1), (128, 16384, 1), (512, 4096, 1), (512, 2048, 1)]
tot = 0.0
for S, Q, K in shapes:
    sup = torch.rand(B, S, 3, device="cuda")
    qry = torch.rand(B, Q, 3, device="cuda")
    for _ in range(3):
        ops.knn_jobs([(sup, qry, K)], B)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    a.record()
    for _ in range(n):
        ops.knn_jobs([(sup, qry, K)], B)
    b.record()
    torch.cuda.synchronize()
    us = a.elapsed_time(b) / n * 1e3
    tot += us
    print("S=%6d Q=%6d K=%2d: %8.1f us  %7.3f Tpair/s" % (S, Q, K, us, B * S * Q / us / 1e6), flush=True)
print("sum %.1f us" % tot)
