"""Where do the training step's library (ATen / Tensile / MIOpen) launches come from?  One profiled step (torch.profiler, with_stack);
every device kernel that is not from libgdm_hip.so is attributed to the innermost frame of THIS package on the stack of the operator
that launched it.  Prints (kernel family, package frame, launches, us).  Development aid."""
import os, sys, collections, re
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from geometric_aware_dense_matching_amd import train_lm, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

B, N, M = 24, 4096, 4096
dev = torch.device("cuda", 0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=True)
ds = train_lm.SyntheticCrops(B, N, M, seed=0)
cu = train_lm.to_device(torch.utils.data.default_collate([ds[i] for i in range(B)]), dev)


def step():
    out, _ = train_lm.model_fn_dec(model, cu, dev)
    out["loss"].backward()
    opt.step(); opt.zero_grad()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    step()
    torch.cuda.synchronize()

PKG = "geometric_aware_dense_matching_amd"
agg = collections.defaultdict(lambda: [0, 0.0])
shapes = collections.defaultdict(collections.Counter)
for ev in prof.events():
    kernels = getattr(ev, "kernels", None)
    if not kernels:
        continue
    for k in kernels:
        name = k.name
        if "anonymous namespace" in name:
            continue                                   # own kernels
        fam = re.sub(r"<.*", "", name.replace("void ", ""))[:60]
        frame = "?"
        for fr in (ev.stack or []):
            if PKG in fr or "tools/" in fr:
                frame = fr.split(PKG + "/")[-1][:90]
                break
        if frame == "?":                                # backward: the autograd node that ran the operator
            par, chain = ev.cpu_parent, []
            while par is not None:
                chain.append(par.name)
                par = par.cpu_parent
            node = [c for c in chain if "Backward" in c or "evaluate_function" in c]
            frame = "(bwd) " + (node[0].replace("autograd::engine::evaluate_function: ", "")[:70] if node else (chain[-1][:70] if chain else ev.name[:60]))
        key = (fam, ev.name[:40], frame)
        agg[key][0] += 1
        agg[key][1] += k.duration
        shapes[key][str(ev.input_shapes)[:100]] += 1
tot = sum(v[1] for v in agg.values())
print("library kernels in one step: %d launches, %.2f ms" % (sum(v[0] for v in agg.values()), tot / 1e3))
for key, (cnt, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:70]:
    print("%4d %8.1f us  %-44s %-28s %s" % (cnt, us, key[0][:44], key[1][:28], key[2]))
    for shp, c in shapes[key].most_common(2):
        print("                  %3d x %s" % (c, shp))
