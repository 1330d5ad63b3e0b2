"""Which side-stream feature breaks hipGraph capture of the step (development aid): python tools/diag_graph_streams.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import infer, synthetic, settings
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
M, N, B = 512, 1024, 2
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).cuda().eval()
b = synthetic.make_batch(seed=1, batch=B, n_points=N)
d = {k: torch.from_numpy(b[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
gp = infer.GraphedPipeline(model, d, with_pose=False)
out = gp(d)
torch.cuda.synchronize()
print("capture + replay ok, side streams =", settings.USE_SIDE_STREAMS, float(out["rgbd"].abs().sum()))
