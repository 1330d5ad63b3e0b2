"""What the matrix pipe sustains on this box: register-only MFMA loops and LDS-fed ones at 1 / 2 / 4 ds_read_b128 per 12 MFMAs
(include/gdm.h gdm_mfma_probe_hip, gdm_mfma_probe_lds_hip).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import _lib, ops
L = _lib.lib()
sink = torch.zeros(4, device="cuda")
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
blocks = 256
for chain in (1, 3):
    it = 3000
    ms = t(lambda: _lib.check(L.gdm_mfma_probe_hip(blocks, it, chain, sink.data_ptr(), ops._stream()), "p"))
    print("registers only, chain %d: %.3f ms  %.0f TFLOP/s" % (chain, ms, blocks * 8 * it * 8 * 16384.0 / ms / 1e9))
for rpu in (1, 2, 4):
    it = 2000
    ms = t(lambda: _lib.check(L.gdm_mfma_probe_lds_hip(blocks, it, rpu, sink.data_ptr(), ops._stream()), "p"))
    print("LDS-fed, %d ds_read_b128 per 12 MFMAs: %.3f ms  %.0f TFLOP/s" % (rpu, ms, blocks * 8 * it * 12 * 16384.0 / ms / 1e9))
