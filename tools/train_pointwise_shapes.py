"""Which per-point layers does one training step run, and how long does each take?  Wraps ops.pointwise with HIP events for ONE
step (after warm-up) and prints the distinct (B, n, K, Cout, weight layout, segments) with call counts and times, beside the bytes a
perfect kernel would move.  Development aid."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import train_lm, synthetic, ops
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
log = collections.OrderedDict()
orig = ops.pointwise


def timed(segs, wt, *a, **kw):
    sl = segs if isinstance(segs, list) else [segs]
    first = sl[0] if torch.is_tensor(sl[0]) else sl[0][0]
    rm = kw.get("w_rowmajor", False)
    K, Cout = (wt.shape[1], wt.shape[0]) if rm else wt.shape
    desc = tuple(("idx" if not torch.is_tensor(s) else "x", (s if torch.is_tensor(s) else s[0]).shape[1]) for s in sl)
    n = (sl[0] if torch.is_tensor(sl[0]) else sl[0][1]).reshape(first.shape[0], -1).shape[1] if not torch.is_tensor(sl[0]) else sl[0].reshape(first.shape[0], first.shape[1], -1).shape[2]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    y = orig(segs, wt, *a, **kw)
    e1.record()
    torch.cuda.synchronize()
    key = (first.shape[0], n, K, Cout, "rowmajor" if rm else "transposed", desc, bool(kw.get("point_major", False)))
    log.setdefault(key, []).append(e0.elapsed_time(e1) * 1e3)
    return y


ops.pointwise = timed
step()
torch.cuda.synchronize()
ops.pointwise = orig
tot = 0.0
print("%5s %7s %5s %5s %-10s %5s %9s %9s  %s" % ("B", "n", "K", "Cout", "W", "calls", "us/call", "ideal us", "segments"))
for k, v in sorted(log.items(), key=lambda kv: -sum(kv[1])):
    b, n, K, Co, lay, desc, pm = k
    ideal = b * n * (K + Co) * 4 / 6.0e12 * 1e6
    tot += sum(v)
    print("%5d %7d %5d %5d %-10s %5d %9.1f %9.1f  %s%s" % (b, n, K, Co, lay, len(v), sum(v) / len(v), ideal, desc, " point-major" if pm else ""))
print("total %.1f us in %d calls" % (tot, sum(len(v) for v in log.values())))
