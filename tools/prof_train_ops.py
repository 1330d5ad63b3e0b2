"""Operator-level view of one training step (torch.profiler, shapes recorded): which aten ops / shapes own the device time that
the kernel trace attributes to anonymous elementwise / copy kernels.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from geometric_aware_dense_matching_amd import train_lm, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

B = int(os.environ.get("B", 24)); N = int(os.environ.get("N", 4096)); M = int(os.environ.get("M", 4096))
dev = torch.device("cuda", 0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4)
ds = train_lm.SyntheticCrops(B, N, M, seed=0)
cu = train_lm.to_device(torch.utils.data.default_collate([ds[i] for i in range(B)]), dev)
def step():
    out, _ = train_lm.model_fn_dec(model, cu, dev)
    out["loss"].backward()
    opt.step(); opt.zero_grad()
for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_input_shape=True).table(sort_by="self_cuda_time_total", row_limit=int(os.environ.get("ROWS", 60)),
                                                         max_name_column_width=40, max_shapes_column_width=90))
