"""Where the time of the MFMA weight-gradient path goes (pack x, pack go, GEMM, sum of parts) against torch's convolution backward:
    python tools/bench_wgrad.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from geometric_aware_dense_matching_amd import _lib, ops


def timed(fn, n=10):
    for _ in range(2):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    L = _lib.lib()
    dev = torch.device("cuda")
    for B, Cin, Cout, H, W in ((24, 512, 512, 32, 32), (24, 256, 256, 32, 32), (24, 256, 512, 32, 32)):
        x = torch.randn(B, Cin, H, W, device=dev)
        go = torch.randn(B, Cout, H, W, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev)
        xpk = torch.empty(L.gdm_wgrad_x_bytes(B, Cin, H, W), dtype=torch.uint8, device=dev)
        gpk = torch.empty(L.gdm_wgrad_go_bytes(B, Cout, H, W), dtype=torch.uint8, device=dev)
        s = ops._stream()
        t_px = timed(lambda: L.gdm_wgrad_pack_x_hip(x.data_ptr(), B, Cin, H, W, xpk.data_ptr(), s))
        t_pg = timed(lambda: L.gdm_wgrad_pack_go_hip(go.data_ptr(), B, Cout, H, W, gpk.data_ptr(), s))
        t_all = timed(lambda: ops.conv3x3_wgrad(x, go))
        t_ref = timed(lambda: torch.ops.aten.convolution_backward(go, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        fl = 2.0 * 9 * Cin * Cout * B * H * W
        print("B=%d %d->%d @%dx%d: pack x %.0f us (%.0f MB), pack go %.0f us, whole own path %.0f us (%.1f TF/s algorithmic), torch %.0f us" %
              (B, Cin, Cout, H, W, t_px, xpk.numel() / 1e6, t_pg, t_all, fl / t_all / 1e6, t_ref))


if __name__ == "__main__":
    main()
