"""Training-step timing (fwd + loss + bwd + Adam) at the reference's default training shape
(config/lmo_cfg.py:95-98,119: N=4096, M=4096, batch 24 per GPU).  Development aid, not the headline bench."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_aware_dense_matching_amd import train_lm, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

B = int(os.environ.get("B", 24)); N = int(os.environ.get("N", 4096)); M = int(os.environ.get("M", 4096))
torch.backends.cudnn.benchmark = os.environ.get("FIND", "0") == "1"
dev = torch.device("cuda", 0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).train()
opt = torch.optim.Adam(model.parameters(), lr=1e-4, fused=os.environ.get("FUSED_ADAM", "1") == "1")
ds = train_lm.SyntheticCrops(B, N, M, seed=0)
batch = torch.utils.data.default_collate([ds[i] for i in range(B)])
cu = train_lm.to_device(batch, dev)
GRAPH = os.environ.get("GRAPH", "0") == "1"
if GRAPH:
    from geometric_aware_dense_matching_amd.train_graph import GraphedTrainStep
    gstep = GraphedTrainStep(model, opt, dev)
def step():
    if GRAPH:
        return gstep.step(batch)
    out, _ = train_lm.model_fn_dec(model, cu, dev)
    out["loss"].backward()
    opt.step(); opt.zero_grad()
    return out
for i in range(5 if GRAPH else 3):
    t1 = time.perf_counter(); out = step(); torch.cuda.synchronize(); print('warmup', i, '%.2f s' % (time.perf_counter() - t1), flush=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
K = 5
for _ in range(K):
    out = step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K
print(("graphed " if GRAPH else "eager ") + "train step B=%d N=%d M=%d: %.1f ms/step = %.1f crops/s; loss %.4f seg %.4f match %.4f; peak mem %.1f GB" %
      (B, N, M, dt * 1e3, B / dt, out["loss"].item(), out["seg_loss"].item(), float(out["match_loss"]), torch.cuda.max_memory_allocated() / 2**30))
