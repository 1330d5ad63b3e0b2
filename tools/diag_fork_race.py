"""Forked hipGraph replays back to back vs the single-stream eager step: which pyramid entries differ, and what the distances say.
Development aid (round 3): python tools/diag_fork_race.py [--rounds 30] [--burst 5] [--parts mesh,point,pyr] [--pyr-only]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops, pyramid, settings, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=30)
ap.add_argument("--burst", type=int, default=5)
ap.add_argument("--parts", default="mesh,point,pyr")
ap.add_argument("--pyr-only", action="store_true", help="the step is the pyramid alone (no model)")
ap.add_argument("--sync-between", action="store_true")
args = ap.parse_args()
B, N, M = 16, 2048, 8192
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M))
tmpl = {k: v for k, v in model.state_dict().items()
        if not k.startswith("model_emb.mesh_graph") and k not in ("model_emb.xyz", "model_emb.const_one")}
model.load_state_dict(synthetic.synthetic_state_dict(tmpl, seed=0), strict=False)
model = model.to(dev).eval()
batch = synthetic.make_batch(seed=100, batch=B, n_points=N)
inputs = {k: torch.from_numpy(batch[k]).to(dev) for k in ("rgb", "cld_rgb_nrm", "choose")}
dpt_xyz = torch.from_numpy(batch["dpt_xyz"]).to(dev)
cld = pyramid.cloud_from_inputs(inputs["cld_rgb_nrm"])


root_buf = torch.zeros(64, device=dev)


def step():
    if os.environ.get("DIAG_SINGLE_ROOT") == "1":
        root_buf.add_(1.0)                          # one kernel on the launch stream in front of every fork: the graph has ONE root
    pyr = pyramid.build_pyramid(cld, dpt_xyz, overlap=True)
    out = {k: v for k, v in pyr.items() if torch.is_tensor(v)}
    for i, t in enumerate(pyr.get(pyramid.KEEP, ())):
        out["keep%d" % i] = t                      # strided grids, sub-clouds, kNN workspace: compared like the results
    if args.pyr_only:
        pyramid.wait_ready(pyr)
        return out
    d = dict(inputs)
    d.update(pyr)
    ep = model(d)
    out.update(rgbd=ep["rgbd"], seg=ep["seg"])
    return out


with torch.no_grad():
    settings.USE_SIDE_STREAMS = False
    settings.USE_SIDE_STREAMS = True
    settings.SIDE_PARTS = args.parts.split(",")
    ref = {k: v.clone() for k, v in step().items()}          # eager, forks on: same values as forks off (tests), and it carries KEEP
    torch.cuda.synchronize()
    pool = ops.BufferPool()
    with ops.buffer_pool(pool):
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = step()
    nbad = 0
    for r in range(args.rounds):
        for _ in range(args.burst):
            g.replay()
            if args.sync_between:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        for k in ref:
            if not torch.equal(ref[k], out[k]):
                nbad += 1
                pos = (ref[k] != out[k]).nonzero()
                print("round %d: %s differs in %d entries" % (r, k, pos.shape[0]))
                for p in pos[:3].tolist():
                    print("   at %s: want %s got %s" % (p, ref[k][tuple(p)].item(), out[k][tuple(p)].item()))
    print("rounds %d burst %d parts %s pyr_only %s sync_between %s: %d differing arrays" % (args.rounds, args.burst, args.parts, args.pyr_only, args.sync_between, nbad))
