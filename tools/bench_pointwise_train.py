"""Per-point layers at training sizes (B = 24; the shapes tools/train_pointwise_shapes.py lists): us per call, back to back, against the
time one read of x and one write of y would take at 6 TB/s.  GDM_PW_CW=0 selects the one-wave-per-channel-group form.  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops

SHAPES = [(24, 16384, 576, 64, False), (24, 4096, 576, 256, False), (24, 16384, 64, 64, True), (24, 16384, 64, 64, False),
          (24, 16384, 128, 64, True), (24, 65536, 32, 32, False), (24, 65536, 32, 32, True), (24, 4096, 64, 128, False),
          (24, 4096, 128, 64, True), (24, 65536, 16, 16, True), (24, 4096, 64, 64, True), (24, 16384, 32, 32, True)]
for B, n, K, Co, rm in SHAPES:
    x = torch.randn(B, K, n, device="cuda")
    w = torch.randn(Co, K, device="cuda") if rm else torch.randn(K, Co, device="cuda")
    y = torch.empty(B, Co, n, device="cuda")
    for _ in range(3):
        ops.pointwise([x], w, out=y, w_rowmajor=rm)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    R = 20
    e0.record()
    for _ in range(R):
        ops.pointwise([x], w, out=y, w_rowmajor=rm)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / R
    ideal = B * n * (K + Co) * 4 / 6.0e12 * 1e6
    print("B %3d n %6d K %4d Cout %4d %-10s %8.1f us  (one pass over x and y at 6 TB/s: %6.1f us, %.2f)" %
          (B, n, K, Co, "rowmajor" if rm else "transposed", us, ideal, ideal / us), flush=True)
    del x, w, y
