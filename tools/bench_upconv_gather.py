import sys, torch
sys.path.insert(0, '/root/repo')
from geometric_aware_dense_matching_amd import ops
for (B, C, H, packed) in ((16, 256, 32, True), (16, 64, 64, False), (16, 64, 64, True)):
    z = torch.randn(B, 9 * C, H, H, device="cuda")
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    for _ in range(3):
        ops.upconv3x3_gather(z, sc, sh, C, (2 * H, 2 * H), 2, 0.25, packed=packed)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.upconv3x3_gather(z, sc, sh, C, (2 * H, 2 * H), 2, 0.25, packed=packed)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 50
    byt = z.numel() * 4 + B * C * 4 * H * H * 4 * (2 if packed else 1)
    print("B %d C %d %d -> %d packed %s: %.1f us  (%.0f MB -> %.2f TB/s)" % (B, C, H, 2 * H, packed, us, byt / 1e6, byt / us / 1e6))
