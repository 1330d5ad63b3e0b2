"""ORACLE -- test infrastructure only; never imported by the product package.

CPU restatement of the neighbour-pyramid builder of the reference loader
(/root/reference/datasets/lm/linemod_pbr.py:515-569; the ycbv loader's copy is
ycbv_pbr.py:541-574).  22 exact-kNN calls per crop produce 30 index / xyz arrays.

`knn` is injectable so that the same restatement can be driven by the real reference
kNN (oracle.knn.ref_knn_batch) when generating golden vectors, or by the C restatement.
"""
import numpy as np

from . import knn as _knn

RGB_DS_SR = [4, 8, 8, 8]        # linemod_pbr.py:529
PCLD_SUB_SR = [4, 4, 4, 4]      # linemod_pbr.py:531
RGB_UP_SR = [4, 2, 2]           # linemod_pbr.py:556
N_DS, N_UP, K_NEI = 4, 3, 16


def strided_xyz_grids(dpt_xyz, S):
    """linemod_pbr.py:515-527: sr2dptxyz[2^i] = xyz map sampled every 2^i pixels, [h*w,3]."""
    full = dpt_xyz.transpose(2, 0, 1)                      # c,h,w   (:515)
    lst = [full]
    for i in range(3):
        sc = 2 ** (i + 1)
        nh = nw = S // sc
        ys, xs = np.mgrid[:nh, :nw]
        lst.append(full[:, ys * sc, xs * sc])              # (:522)
    return {2 ** ii: np.ascontiguousarray(item.reshape(3, -1).transpose(1, 0)) for ii, item in enumerate(lst)}


def build_pyramid(cld, dpt_xyz, knn_search=None):
    """cld f32[N,3] (sampled scene points), dpt_xyz f32[S,S,3] -> dict of 30 arrays."""
    if knn_search is None:
        knn_search = _knn.knn_search
    S = dpt_xyz.shape[0]
    sr2 = strided_xyz_grids(dpt_xyz.astype(np.float32), S)
    cld = cld.astype(np.float32)
    out = {}
    for i in range(N_DS):                                   # :533-554
        nei = knn_search(cld[None], cld[None], K_NEI).astype(np.int32)[0]
        n_sub = cld.shape[0] // PCLD_SUB_SR[i]
        sub = cld[:n_sub]                                   # "random" sampling == prefix slice (:538)
        out["cld_xyz%d" % i] = cld.copy()
        out["cld_nei_idx%d" % i] = nei.copy()
        out["cld_sub_idx%d" % i] = nei[:n_sub].copy()
        out["cld_interp_idx%d" % i] = knn_search(sub[None], cld[None], 1).astype(np.int32)[0]
        px = sr2[RGB_DS_SR[i]]
        out["r2p_ds_nei_idx%d" % i] = knn_search(px[None], sub[None], K_NEI).astype(np.int32)[0]
        out["p2r_ds_nei_idx%d" % i] = knn_search(sub[None], px[None], 1).astype(np.int32)[0]
        cld = sub
    for i in range(N_UP):                                   # :556-568
        pts = out["cld_xyz%d" % (N_DS - i - 1)]
        px = sr2[RGB_UP_SR[i]]
        out["r2p_up_nei_idx%d" % i] = knn_search(px[None], pts[None], K_NEI).astype(np.int32)[0]
        out["p2r_up_nei_idx%d" % i] = knn_search(pts[None], px[None], 1).astype(np.int32)[0]
    return out


def ref_knn_search(support, query, k):
    """helper_tool.py:160-170 over the REAL reference kNN (oracle/_ref)."""
    return _knn.ref_knn_batch(support, query, k, omp=True).astype(np.int32)
