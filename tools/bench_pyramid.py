"""Times the neighbour pyramid (22 kNN searches per crop, two launches per batch) at the bench shape with HIP events.
Development aid: python tools/bench_pyramid.py [B] [N]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import pyramid, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
batch = synthetic.make_batch(seed=100, batch=B, n_points=N)
cld = pyramid.cloud_from_inputs(torch.from_numpy(batch["cld_rgb_nrm"]).cuda())
xyz = torch.from_numpy(batch["dpt_xyz"]).cuda()
for _ in range(200):                       # ~60 ms: a chip coming out of idle runs its first tens of milliseconds several times slower
    pyramid.build_pyramid(cld, xyz)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
a.record()
for _ in range(n):
    pyramid.build_pyramid(cld, xyz)
b.record()
torch.cuda.synchronize()
print("pyramid B=%d N=%d: %.3f ms per batch" % (B, N, a.elapsed_time(b) / n))
