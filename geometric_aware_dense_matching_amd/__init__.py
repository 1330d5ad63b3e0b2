"""MI355X-native implementation of the geoMatch dense-correspondence hot path.

Scope: the exact-kNN neighbour pyramid, the RandLA / fusion gather ops,
the GeoMatch forward (reference surface `models.geoMatch.GeoMatch(cfg, cls_id).forward(inputs)`)
and the N x M descriptor matching, as hand-written HIP kernels for gfx950 behind a C ABI
(include/gdm.h, libgdm_hip.so) with a thin PyTorch-ROCm host layer.  No CPU fallback.
"""
__version__ = "0.1.0"
