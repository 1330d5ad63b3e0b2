"""Where does a hipGraph replay of the bench step differ from the eager step?  (development aid, round 3)

Runs the step of bench.py (neighbour pyramid + GeoMatch.forward + matching) at the bench shape with the side-stream forks
off and with each fork alone, and compares -- bit for bit -- every intermediate (30 pyramid arrays, rgbd, seg, mesh, arg-max
indices, maxima) between eager steps, between graph replays, and between the two.

    python tools/diag_graph_check.py [--batch 16] [--replays 10] [--configs off,pyr,mesh,point,all]
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from geometric_aware_dense_matching_amd import ops, pyramid, settings, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--npoints", type=int, default=2048)
    ap.add_argument("--mesh", type=int, default=8192)
    ap.add_argument("--replays", type=int, default=10)
    ap.add_argument("--configs", default="off,pyr,mesh,point,all")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    B, N, M = args.batch, args.npoints, args.mesh
    dev = torch.device("cuda", 0)
    torch.backends.cudnn.benchmark = True
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

    def step():
        pyr = pyramid.build_pyramid(cld, dpt_xyz, overlap=True)
        d = dict(inputs)
        d.update(pyr)
        ep = model(d)
        mask, count = ops.seg_mask(ep["seg"])
        srows = ops.match_pack(ep["rgbd"], ops.MATCH_BF16X3)
        mrows = ops.match_pack(ep["mesh"][0], ops.MATCH_BF16X3)
        bi, bs = ops.match_packed(srows, mrows, B, N, M, ops.MATCH_BF16X3)
        out = {k: v for k, v in pyr.items() if torch.is_tensor(v)}
        out.update(rgbd=ep["rgbd"], seg=ep["seg"], mesh=ep["mesh"], mask=mask, best_idx=bi, best_sim=bs)
        return out

    def snap(o):
        return {k: v.clone() for k, v in o.items()}

    def diff(a, b):
        bad = {}
        for k in a:
            if not torch.equal(a[k], b[k]):
                n = int((a[k] != b[k]).sum())
                mx = float((a[k].double() - b[k].double()).abs().max())
                bad[k] = {"n_diff": n, "of": a[k].numel(), "max_abs": mx}
        return bad

    report = {}
    with torch.no_grad():
        for cfg in args.configs.split(","):
            settings.USE_SIDE_STREAMS = cfg != "off"
            settings.SIDE_PARTS = ["mesh", "point", "pyr", "psp"] if cfg == "all" else cfg.split("+")
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            e1 = snap(step())
            torch.cuda.synchronize()
            e2 = snap(step())
            torch.cuda.synchronize()
            r = {"eager_vs_eager": diff(e1, e2)}
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                gout = step()
            torch.cuda.synchronize()
            worst = {}
            first = None
            nbad = 0
            for i in range(args.replays):
                g.replay()
                torch.cuda.synchronize()
                s = snap(gout)
                d = diff(e1, s)
                if d:
                    nbad += 1
                    for k, v in d.items():
                        if k not in worst or v["n_diff"] > worst[k]["n_diff"]:
                            worst[k] = v
                if first is None:
                    first = s
                else:
                    dd = diff(first, s)
                    if dd:
                        r.setdefault("replay_vs_replay", {}).update(dd)
            # back-to-back replays without a host sync in between (the timed loop's form), then compare the last one
            for i in range(args.replays):
                g.replay()
            torch.cuda.synchronize()
            r["replay_vs_eager_bad_replays"] = nbad
            r["replay_vs_eager_worst"] = worst
            r["back_to_back_replays_vs_eager"] = diff(e1, snap(gout))
            e3 = snap(step())
            torch.cuda.synchronize()
            r["eager_after_graph_vs_eager"] = diff(e1, e3)
            report[cfg] = r
            print(cfg, json.dumps(r), flush=True)
            del g, gout
    if args.out:
        with open(args.out, "w") as f:
            json.dump(report, f, indent=1)


if __name__ == "__main__":
    main()
