"""The four adaptive average pools of the pyramid-pooling module alone (16 x 1024 planes of 32 x 32).  Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from geometric_aware_dense_matching_amd import ops
x = torch.randn(16, 1024, 32, 32, device="cuda")
from geometric_aware_dense_matching_amd import _lib
o = ops.psp_pools(x)
L = _lib.lib()
call = lambda: L.gdm_psp_pools_hip(x.data_ptr(), 16 * 1024, 32, 32, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), None)
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(30): call()                                        # the C entry point on preallocated outputs: no host-side allocation in the loop
e1.record(); torch.cuda.synchronize()
ref = [torch.nn.functional.adaptive_avg_pool2d(x, s) for s in (1, 2, 3, 6)]
err = max((a.reshape(-1) - b.reshape(-1)).abs().max().item() for a, b in zip(o, ref))
print("psp_pools: %.1f us per launch (67 MB read: %.2f TB/s); max |err| vs adaptive_avg_pool2d %.1e" % (e0.elapsed_time(e1) * 1e3 / 30, 67.1 / (e0.elapsed_time(e1) / 30) / 1e3 * 1e0, err))
import ctypes
old = "tools/micro/variants/libgdm_oldpools.so"
if os.path.exists(old):
    l = ctypes.CDLL(old)
    f = l.gdm_psp_pools_hip; f.restype = ctypes.c_int; f.argtypes = _lib.SIGNATURES["gdm_psp_pools_hip"][1]
    call2 = lambda: f(x.data_ptr(), 16 * 1024, 32, 32, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), None)
    for _ in range(3): call2()
    torch.cuda.synchronize(); e0.record()
    for _ in range(30): call2()
    e1.record(); torch.cuda.synchronize()
    print("previous build: %.1f us per launch" % (e0.elapsed_time(e1) * 1e3 / 30))
