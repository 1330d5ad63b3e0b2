"""`final` (Conv2d(64, 64, 1) + LogSoftmax) at the step's size, 16 x 64 x 128 x 128.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
x = torch.randn(16, 64, 128, 128, device="cuda"); w = torch.randn(64, 64, 1, 1, device="cuda") / 8; b = torch.randn(64, device="cuda")
for _ in range(3): y = ops.conv1x1_logsoftmax(x, w, b)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): y = ops.conv1x1_logsoftmax(x, w, b)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / 30
print("final stage: %.1f us per launch (134 MB in + out: %.2f TB/s)" % (us, 134.2 / us))
