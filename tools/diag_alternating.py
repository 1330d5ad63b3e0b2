"""Forked hipGraph replays with ALTERNATING inputs, back to back, every replay checked against the eager step of its batch.
Decides whether anything in the forked step depends on an ordering that constant-input replays cannot see (an overlap between
consecutive replays, or a missing edge inside the graph: both are invisible when producer and consumer see the same bytes twice).
    python tools/diag_alternating.py [--rounds 40] [--burst 8] [--parts mesh,point,pyr]
(The GDM_PYR_SINGLE / GDM_PYR_DUMMY / GDM_KNN1_PAIR hooks this tool was used with in round 4 were removed with the kernel: commit 5c3353f holds them.)"""
import argparse, collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import infer, ops, settings, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

ap = argparse.ArgumentParser()
ap.add_argument("--rounds", type=int, default=40)
ap.add_argument("--burst", type=int, default=8)
ap.add_argument("--parts", default="mesh,point,pyr")
ap.add_argument("--same", action="store_true", help="control: the same batch every replay (what round 3's stress test did)")
args = ap.parse_args()
B, N, M = 16, 2048, 8192
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M))
tmpl = {k: v for k, v in model.state_dict().items()
        if not k.startswith("model_emb.mesh_graph") and k not in ("model_emb.xyz", "model_emb.const_one")}
model.load_state_dict(synthetic.synthetic_state_dict(tmpl, seed=0), strict=False)
model = model.to(dev).eval()
keys = ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")
batches = []
for seed in (100, 733):
    b = synthetic.make_batch(seed=seed, batch=B, n_points=N)
    batches.append({k: torch.from_numpy(b[k]).to(dev) for k in keys})
settings.SIDE_PARTS = args.parts.split(",")
with torch.no_grad():
    eager = []
    for d in batches:
        o = infer.pipeline_step(model, d, with_pose=False, keep_pyramid=True)
        eager.append({k: v.clone() for k, v in o.items() if torch.is_tensor(v)})
    torch.cuda.synchronize()
    gp = infer.GraphedPipeline(model, batches[0], with_pose=False, keep_pyramid=True, forked="auto", keep_both=True)
print("form kept by GraphedPipeline:", gp.form, gp.check)
if "forked" not in gp.graphs:
    # the constant-input check already failed: drive the forked capture anyway to see what alternating inputs show
    settings.USE_SIDE_STREAMS = True
    with torch.no_grad():
        gp._capture("forked", 2)
    settings.USE_SIDE_STREAMS = False
g, out = gp.graphs["forked"], gp.outs["forked"]
side = [{k: torch.empty_like(v) for k, v in out.items() if torch.is_tensor(v)} for _ in range(args.burst)]
count = collections.Counter()
nbad = 0
torch.cuda.synchronize()
for r in range(args.rounds):
    for i in range(args.burst):
        src = batches[0 if args.same else i % 2]
        for k, buf in gp.static_in.items():
            buf.copy_(src[k], non_blocking=True)
        g.replay()
        for k, buf in side[i].items():
            buf.copy_(out[k], non_blocking=True)
    torch.cuda.synchronize()
    for i in range(args.burst):
        ok, names = infer.outputs_equal(eager[0 if args.same else i % 2], side[i])
        if not ok:
            nbad += 1
            count.update(names)
            if nbad <= 5:
                for k in names[:4]:
                    a, b = eager[0 if args.same else i % 2][k], side[i][k]
                    pos = (a != b).nonzero()
                    print("round %d replay %d: %s differs in %d entries, first %s want %s got %s" % (
                        r, i, k, pos.shape[0], pos[0].tolist(), a[tuple(pos[0])].item(), b[tuple(pos[0])].item()))
print("rounds %d x burst %d, parts %s, same=%s, PYR_SINGLE=%s KNN1_PAIR=%s: %d of %d replays differ; arrays: %s" % (
    args.rounds, args.burst, args.parts, args.same, os.environ.get("GDM_PYR_SINGLE"), os.environ.get("GDM_KNN1_PAIR"), nbad,
    args.rounds * args.burst, dict(count)))
