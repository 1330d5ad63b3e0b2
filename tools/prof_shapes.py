"""torch.profiler with shapes for one eval forward: which aten GEMM / conv calls remain and how long they take. Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from geometric_aware_dense_matching_amd import pyramid, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
torch.backends.cudnn.benchmark = True
M, N, B = 8192, 2048, 16
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).cuda().eval()
batch = synthetic.make_batch(seed=1, batch=B, n_points=N)
inp = {k: torch.from_numpy(batch[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
def step():
    d = dict(inp); d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))
    with torch.no_grad():
        return model(d)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(); torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    if e.key in ("aten::mm", "aten::bmm", "aten::addmm", "aten::matmul", "aten::conv2d", "aten::miopen_convolution", "aten::convolution", "aten::cat", "aten::linear", "aten::conv1d"):
        t = getattr(e, "device_time_total", None) or getattr(e, "cuda_time_total", 0)
        rows.append((t, e.key, e.count, str(e.input_shapes)[:150]))
for t, k, c, s in sorted(rows, reverse=True)[:40]:
    print("%9.1f us  %-26s x%-3d %s" % (t, k, c, s))
