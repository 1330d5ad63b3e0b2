"""ORACLE -- test infrastructure only; never imported by the product package.

numpy restatements of the `lib/pointops` wrapper contracts (/root/reference/lib/pointops/functions/pointops.py; the CUDA kernels
behind them are absent from the reference, so what is restated is the wrapper's documented input -> output relation and, where
the wrapper holds a torch fallback (KNNQueryNaive :395-432, KNNQueryExclude :496-533, pairwise_distances :375-392), that code).
Parity is therefore "API contract", not arithmetic of a reference kernel."""
import numpy as np


def d2(a, b):
    """(n,3),(m,3) -> (n,m) squared distances, fp32, ((dx*dx)+dy*dy)+dz*dz like the kNN oracle."""
    d = a[:, None, :].astype(np.float32) - b[None, :, :].astype(np.float32)
    return ((d[..., 0] * d[..., 0]) + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]


def knn(xyz, new_xyz, k):
    """idx (b,m,k) nearest first, ties by index (stable sort) -- KNNQueryNaive's sort on exact fp32 distances."""
    out = np.zeros((xyz.shape[0], new_xyz.shape[1], k), np.int32)
    for b in range(xyz.shape[0]):
        out[b] = np.argsort(d2(new_xyz[b], xyz[b]), axis=1, kind="stable")[:, :k]
    return out


def nearestneighbor(unknown, known):
    idx = knn(known, unknown, 3)
    dist = np.stack([np.sqrt(np.take_along_axis(d2(unknown[b], known[b]), idx[b].astype(np.int64), axis=1)) for b in range(len(idx))])
    return dist.astype(np.float32), idx


def interpolation(features, idx, weight):
    b, c, m = features.shape
    out = np.zeros((b, c, idx.shape[1]), np.float32)
    for k in range(3):
        out += np.take_along_axis(features, np.broadcast_to(idx[:, None, :, k], (b, c, idx.shape[1])).astype(np.int64), axis=2) * weight[:, None, :, k]
    return out


def interpolation_backward(grad_out, idx, weight, m):
    b, c, n = grad_out.shape
    g = np.zeros((b, c, m), np.float64)
    for bb in range(b):
        for k in range(3):
            np.add.at(g[bb], (slice(None), idx[bb, :, k]), grad_out[bb] * weight[bb, None, :, k])
    return g.astype(np.float32)


def ballquery(radius, nsample, xyz, new_xyz):
    b, m = new_xyz.shape[:2]
    out = np.zeros((b, m, nsample), np.int32)
    for bb in range(b):
        inside = d2(new_xyz[bb], xyz[bb]) < np.float32(radius) * np.float32(radius)
        for j in range(m):
            hit = np.nonzero(inside[j])[0][:nsample]
            if len(hit):
                out[bb, j] = hit[0]
                out[bb, j, :len(hit)] = hit
    return out


def labelstat_ballrange(radius, xyz, new_xyz, label_stat):
    b, m = new_xyz.shape[:2]
    out = np.zeros((b, m, label_stat.shape[2]), np.int32)
    for bb in range(b):
        inside = d2(new_xyz[bb], xyz[bb]) < np.float32(radius) * np.float32(radius)
        out[bb] = inside.astype(np.int64) @ label_stat[bb].astype(np.int64)
    return out


def labelstat_idx(label_stat, idx):
    return np.stack([label_stat[b][idx[b]].sum(axis=1) for b in range(len(idx))]).astype(np.int32)
