"""PSPUpsample(64 -> 64) kernel alone at the last up stage's shape (batch 16, 128^2 -> 256^2).  Development aid.
(The z-gather form it could be compared with, GDM_UPCONV_FUSED64=gather, was removed in round 4.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from geometric_aware_dense_matching_amd import ops

dev = torch.device("cuda", 0)
torch.manual_seed(0)
B, H = int(os.environ.get("B", 16)), int(os.environ.get("H", 128))
x = torch.randn(B, 64, H, H, device=dev)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev)
wk = ops.upconv_fused64_pack_weight(w)
out = ops.upconv_fused64(x, wk, sc, sh, (2 * H, 2 * H), 2, 0.25)
ref = F.conv2d(F.interpolate(x[:2].double(), size=(2 * H, 2 * H), mode="bilinear", align_corners=True), w.double(), padding=1)
ref = ref * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
ref = torch.where(ref > 0, ref, 0.25 * ref)
print("max |err| vs fp64: %.3e (max |ref| %.2f)" % ((out[:2].double() - ref).abs().max().item(), ref.abs().max().item()))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    ops.upconv_fused64(x, wk, sc, sh, (2 * H, 2 * H), 2, 0.25)
torch.cuda.synchronize(); a.record()
n = 20
for _ in range(n):
    ops.upconv_fused64(x, wk, sc, sh, (2 * H, 2 * H), 2, 0.25)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / n * 1e3
fl = 2.0 * B * 4 * H * H * 64 * 64 * 9
print("form %s: %.1f us; %.0f GB/s of in+out; %.0f TF/s bf16 issued at output resolution" %
      ("tile", us, (B * 64 * H * H * 4 * 5) / us / 1e3, 3 * fl / us / 1e6))
