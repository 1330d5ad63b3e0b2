"""Where the stem kernel's time goes: builds of gdm_stem.hip with -DGDM_STEM_ABL=1 (one MFMA product of three), =4 (no epilogue), =8 (patch fill
only) as tools/micro/variants/libgdm_stem{1,4,8}.so, loaded in turn.  Round 4: shipped 50.6 us = patch fill 13.4 + implicit GEMM 13.4 +
BN / ReLU / pool / stores 23.8; after the epilogue's masks were hoisted 42.8 us (round 3: 64 us with the fill as a load -> store loop).
Build the variants first:  cd csrc; for v in 1 4 8; do hipcc ... -DGDM_STEM_ABL=$v -c gdm_stem.hip -o /tmp/stem_$v.o && hipcc -shared ... ; done"""
import ctypes, os, sys, torch
sys.path.insert(0, '/root/repo')
from geometric_aware_dense_matching_amd import _lib, ops
x = torch.randn(16, 3, 256, 256, device="cuda"); w = torch.randn(64, 3, 7, 7, device="cuda") * 0.05
sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda")
wpk = ops.stem_pack_weight(w)
def run(tag):
    for _ in range(3): ops.stem(x, wpk, sc, sh)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): ops.stem(x, wpk, sc, sh)
    e1.record(); torch.cuda.synchronize()
    print("%-40s %.1f us" % (tag, e0.elapsed_time(e1) * 1e3 / 30), flush=True)
run("shipped")
for v, what in ((1, "one MFMA product of three"), (4, "no epilogue (BN, pool, stores)"), (8, "patch fill only")):
    l = ctypes.CDLL("/root/repo/tools/micro/variants/libgdm_stem%d.so" % v)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(l, name); fn.restype = res; fn.argtypes = args
    _lib._lib = l
    run(what)
