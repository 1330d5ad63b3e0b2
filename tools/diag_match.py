import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_aware_dense_matching_amd import ops
B, N, M = 16, 2048, 8192
torch.manual_seed(0)
scene = torch.randn(B, 128, N, device="cuda"); model = torch.randn(128, M, device="cuda")
sim = torch.empty(B, N, M, device="cuda")
srows, mrows = ops.match_pack(scene, 0), ops.match_pack(model, 0)
def t(mode, diag):
    os.environ["GDM_MATCH_DIAG"] = diag
    ts = []
    for r in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(5):
            ops.match_packed(srows, mrows, B, N, M, 0, return_sim=(mode == "mat"), sim_out=sim if mode == "mat" else None)
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 5)
    return np.median(ts[1:]) * 1e3
for _ in range(2):
    print("fused %.1f us | materialised %.1f us | stores only (no MFMA) %.1f us" % (t("fused", "0"), t("mat", "0"), t("mat", "1")))
x = torch.empty(268435456, device="cuda")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x.fill_(1.0); torch.cuda.synchronize(); a.record()
for _ in range(5): x.fill_(2.0)
b.record(); torch.cuda.synchronize()
print("torch fill 1.07 GB: %.1f us" % (a.elapsed_time(b) / 5 * 1e3))
