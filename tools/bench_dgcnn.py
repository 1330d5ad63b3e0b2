"""geoMatch_DGCNN variant (BASELINE config 4): eval forward + matching throughput, and where the time goes. Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from geometric_aware_dense_matching_amd import matching, synthetic
from geometric_aware_dense_matching_amd.geoMatch_DGCNN import GeoMatch
B, N, M = int(os.environ.get("B", 16)), 2048, int(os.environ.get("M", 8192))
model = GeoMatch(dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, n_mesh_node=M), 1, model_points=synthetic.make_model_points(1, M)).cuda().eval()
batch = synthetic.make_batch(seed=1, batch=B, n_points=N)
inp = {k: torch.from_numpy(batch[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose")}
def step():
    with torch.no_grad():
        ep = model(inp)
        return matching.match_frames(ep)
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print("DGCNN variant B=%d N=%d M=%d: %.2f ms/step = %.0f crops/s" % (B, N, M, dt * 1e3, B / dt))
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    step(); torch.cuda.synchronize()
rows = sorted(((e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total, e.key, e.count) for e in prof.key_averages()), reverse=True)[:14]
for t, k, c in rows: print("%9.1f us  x%-3d %s" % (t, c, k[:110]))
