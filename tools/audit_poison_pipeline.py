"""Does the headline step read memory nobody wrote?  The eager step and the hipGraph forms with every torch.empty() filled with NaN /
INT_MAX (torch.utils.deterministic.fill_uninitialized_memory) must give the bits of the plain run.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import infer, ops, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

dev = torch.device("cuda", 0)
B, N, M = 16, 2048, 8192
torch.manual_seed(0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).eval()
b = synthetic.make_batch(seed=1, batch=B, n_points=N)
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
with torch.no_grad():
    ref = {k: v.clone() for k, v in infer.pipeline_step(model, d, with_pose=True, keep_pyramid=True).items() if torch.is_tensor(v)}
torch.cuda.synchronize()
torch.use_deterministic_algorithms(True, warn_only=True)
torch.utils.deterministic.fill_uninitialized_memory = True
import warnings; warnings.filterwarnings("ignore")
with torch.no_grad():
    with ops.buffer_pool(ops.BufferPool()):                       # fresh scratch buffers too
        out = infer.pipeline_step(model, d, with_pose=True, keep_pyramid=True)
    torch.cuda.synchronize()
    ok, bad = infer.outputs_equal(ref, out)
    print("eager step with poisoned allocations == plain run:", ok, bad[:5])
    gp = infer.GraphedPipeline(model, d, with_pose=True, keep_pyramid=True)
    o2 = gp(d); torch.cuda.synchronize()
    ok2, bad2 = infer.outputs_equal(ref, o2)
    print("hipGraph (%s form) with poisoned allocations == plain run:" % gp.form, ok2, bad2[:5], gp.check)
