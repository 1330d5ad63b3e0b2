"""Which section of the step breaks hipGraph capture? Development aid."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import matching, pose, pyramid, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
M, N, B = 2048, 1024, 2
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M))
model = model.cuda().eval()
batch = synthetic.make_batch(seed=1, batch=B, n_points=N)
inp = {k: torch.from_numpy(batch[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
state = {}
def s_pyr():
    state["d"] = dict(inp); state["d"].update(pyramid.build_pyramid(pyramid.cloud_from_inputs(inp["cld_rgb_nrm"]), inp["dpt_xyz"]))
def s_fwd():
    state["ep"] = model(state["d"])
def s_emb():
    state["e"] = model.pcd_emb(state["d"])
def s_mesh():
    state["m"] = model.model_emb()
def s_match():
    state["res"] = matching.match_frames(state["ep"])
def s_pose():
    state["rt"] = pose.solve_poses(state["res"], inp["cld_rgb_nrm"], model.model_emb.xyz)
for name, fn in (("pyramid", s_pyr), ("ffb6d", s_emb), ("mesh", s_mesh), ("forward", s_fwd), ("match", s_match), ("pose", s_pose)):
    with torch.no_grad():
        for _ in range(2): fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                fn()
            g.replay(); torch.cuda.synchronize()
            print(name, "OK", flush=True)
        except Exception as e:
            print(name, "FAILED:", str(e).splitlines()[0], flush=True)
            tb = traceback.extract_tb(e.__traceback__)
            for fr in tb[-6:]:
                print("   ", fr.filename.split("/")[-1], fr.lineno, fr.name, flush=True)
            torch.cuda.synchronize()
