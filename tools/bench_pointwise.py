"""Per-shape timing of the per-point 1x1 layer kernel (gdm_pointwise_hip) at the shapes of the eval step (batch 16, N = 2048):
    python tools/bench_pointwise.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from geometric_aware_dense_matching_amd import ops

B = 16
# (points per crop, K segments, Cout, what)
SHAPES = [
    (2048, (9,), 8, "fc0"), (2048, (8,), 16, "L0 mlp1"), (2048, (32, 8), 64, "L0 tail"),
    (512, (64,), 32, "L1 mlp1"), (512, (64, 64), 128, "L1 tail"), (512, (64,), 64, "r2p_pre0"), (512, (64, 64), 64, "r2p_fuse0"),
    (128, (128,), 64, "L2 mlp1"), (128, (128, 128), 256, "L2 tail"), (128, (128,), 128, "r2p_pre1"), (128, (128, 128), 128, "r2p_fuse1"),
    (32, (256,), 128, "L3 mlp1"), (32, (256, 256), 512, "L3 tail"), (32, (512,), 256, "r2p_pre2"), (32, (256, 256), 256, "r2p_fuse2"),
    (32, (256,), 512, "p2r_pre2"), (32, (512,), 512, "p2r_t2"),
    (8, (1024,), 512, "r2p_pre3"), (8, (512, 512), 512, "r2p_fuse3"), (8, (512,), 1024, "p2r_pre3"), (8, (1024,), 1024, "p2r_t3"),
    (32, (512, 512), 256, "dec0"), (128, (256, 256), 128, "dec1"), (512, (128, 128), 64, "dec2"), (2048, (64, 64), 64, "dec3"),
]


def main():
    dev = torch.device("cuda")
    tot = 0.0
    for n, cs, cout, what in SHAPES:
        segs = [torch.randn(B, c, n, device=dev) for c in cs]
        K = sum(cs)
        wt = torch.randn(K, cout, device=dev) / K ** 0.5
        sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
        f = lambda: ops.pointwise(segs, wt, sc, sh, ops.ACT_LEAKY, 0.2)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()               # 50 launches as one graph: the eager loop is bound by the host (~11 us per call)
        with torch.cuda.graph(g):
            for _ in range(50):
                f()
        g.replay()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        g.replay()
        b.record()
        torch.cuda.synchronize()
        us = a.elapsed_time(b) / 50 * 1e3
        fl = 2.0 * B * n * K * cout
        tot += us
        print("%-10s P=%6d K=%5d Cout=%5d  %7.1f us  %6.2f TFLOP/s" % (what, B * n, K, cout, us, fl / us / 1e6))
    print("sum %.1f us" % tot)


if __name__ == "__main__":
    main()
