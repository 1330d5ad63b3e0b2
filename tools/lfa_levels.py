import os, sys
sys.path.insert(0, "/root/repo")
import torch
from geometric_aware_dense_matching_amd import randla, settings
B, K = 16, 16
fused = os.environ.get("FUSED", "1") == "1"
settings.USE_FUSED_LFA = fused
for d_out, n in ((32, 2048), (64, 512), (128, 128), (256, 32)):
    torch.manual_seed(0)
    blk = randla.BuildingBlock(d_out).cuda().eval()
    xyz = torch.randn(B, n, 3, device="cuda"); feat = torch.randn(B, d_out // 2, n, 1, device="cuda")
    idx = torch.randint(0, n, (B, n, K), device="cuda", dtype=torch.int32)
    with torch.no_grad():
        for _ in range(3):
            blk(xyz, feat, idx)
        torch.cuda.synchronize()
        # marker kernel between levels: a fill of distinctive size
        torch.zeros(12345 + d_out, device="cuda")
        for _ in range(5):
            blk(xyz, feat, idx)
        torch.cuda.synchronize()
        torch.zeros(54321 + d_out, device="cuda")
print("done")
