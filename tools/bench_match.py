"""A/B timing of the match kernels (development aid): interleaved rounds in ONE process."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from geometric_aware_dense_matching_amd import ops

B, N, M = int(os.environ.get("B", 16)), 2048, 8192
torch.manual_seed(0)
scene = torch.randn(B, 128, N, device="cuda")
model = torch.randn(128, M, device="cuda")
sim = torch.empty(B, N, M, device="cuda")
flops = 2.0 * B * N * M * 128
mat_bytes = 4.0 * 128 * (B * N + M) + 4.0 * B * N * M
res = {}
for prec, pname in ((0, "bf16x3"), (1, "f32")):
    srows = ops.match_pack(scene, prec)
    mrows = ops.match_pack(model, prec)
    outs = {}
    times = {}
    for rnd in range(5):
        for ver in ("2", "3"):
            os.environ["GDM_MATCH_KERNEL"] = ver
            for mode in ("fused", "mat"):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                a.record()
                for _ in range(5):
                    o = ops.match_packed(srows, mrows, B, N, M, prec, return_sim=(mode == "mat"), sim_out=sim if mode == "mat" else None)
                b.record()
                torch.cuda.synchronize()
                times.setdefault((ver, mode), []).append(a.elapsed_time(b) / 5)
                outs[(ver, mode)] = (o[0].clone(), o[1].clone(), sim[0, :64, :64].clone() if mode == "mat" else None)
    for k in sorted(times):
        ms = np.median(times[k][1:])
        extra = ("%.0f GB/s (%.1f%% of 8 TB/s)" % (mat_bytes / ms / 1e6, mat_bytes / ms / 1e6 / 80)) if k[1] == "mat" else ""
        print("%-7s v%s %-5s %8.1f us  %7.1f TF/s algorithmic  %s" % (pname, k[0], k[1], ms * 1e3, flops / ms / 1e9, extra))
    same_idx = torch.equal(outs[("3", "fused")][0], outs[("2", "fused")][0])
    dv = (outs[("3", "fused")][1] - outs[("2", "fused")][1]).abs().max().item()
    ds = (outs[("3", "mat")][2] - outs[("2", "mat")][2]).abs().max().item()
    print("   v3 vs v2: idx equal=%s  max|dval|=%.2e  max|dsim|=%.2e  fused==mat idx: %s" %
          (same_idx, dv, ds, torch.equal(outs[("2", "fused")][0], outs[("2", "mat")][0])))
