"""The stem kernel alone (conv 7x7 / 2 + BN + ReLU + max-pool, 16 x 3 x 256 x 256 -> 16 x 64 x 64 x 64 + packed operand).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
x = torch.randn(16, 3, 256, 256, device="cuda"); w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
wpk = ops.stem_pack_weight(w)
for _ in range(3): y = ops.stem(x, wpk, sc, sh)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): y = ops.stem(x, wpk, sc, sh)
e1.record(); torch.cuda.synchronize()
ref = torch.nn.functional.max_pool2d(torch.relu(torch.nn.functional.conv2d(x, w, stride=2, padding=3) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1)
print("stem: %.1f us per launch; max |err| vs the fp32 modules %.2e (scale %.1f)" % (e0.elapsed_time(e1) * 1e3 / 30, (y - ref).abs().max().item(), ref.abs().max().item()))
