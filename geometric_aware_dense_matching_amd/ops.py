"""Torch-facing operators over the C ABI of libgdm_hip.so.

PyTorch is plumbing here: it owns device memory and the current HIP stream; every operator
passes raw device pointers + that stream to the hand-written HIP kernels.  There is no CPU
path: CPU tensors raise.  Index tensors are int32 on the device (the reference widens them to
int64 only because torch.gather demands it: /root/reference/train_lm.py:167-168); int64 is
accepted and narrowed.

Operator <-> reference map (file:line under /root/reference):
  knn_batch / knn_jobs   models/RandLA/utils/nearest_neighbors/knn.pyx:71-109, knn_.cxx:104-135
  group_gather           models/RandLA/RandLANet.py:729-738 (+ permute :704-716); pointops.py:151-176
  gather_max             models/ffb6d.py:128-146 random_sample
  gather_nn              models/ffb6d.py:148-163 nearest_interpolation; pointops.py:61-82
  rel_pos_enc            models/RandLA/RandLANet.py:720-727
  att_pool               models/RandLA/RandLANet.py:749-752
  match / seg_mask       evaluator.py:79-93
  ballquery / furthestsampling   lib/pointops/functions/pointops.py:205-219, 40-50
"""
import ctypes
import os

import torch

from . import _lib, settings
from ._lib import KnnJob, check

MATCH_BF16X3 = 0
MATCH_F32 = 1


def _stream():
    return torch.cuda.current_stream().cuda_stream


_side_streams = {}


def side_stream(device, which=0):
    """A per-device side stream (HIP streams let independent branches of the step overlap: most of its ~120 kernels are far too
    small to fill 256 CUs).  Callers fork with side.wait_stream(current), join with current.wait_stream(side)."""
    key = (device.index if device.index is not None else torch.cuda.current_device(), which)
    st = _side_streams.get(key)
    if st is None:
        st = torch.cuda.Stream(device=device)
        _side_streams[key] = st
    return st


def _stream_lane(device):
    """Which ROLE the current stream plays for the scratch-buffer pools: ("side", k) for side stream k of the device, "main" for any
    other stream.  The pools are keyed by role, not by stream handle: a hipGraph capture runs on a stream of its own (torch's capture
    stream), and a pool keyed by handle missed every buffer the eager warm-up had made -- the capture then allocated and ZERO-FILLED
    them again inside the graph (12 fill kernels, 0.13-0.15 ms of every replay).  A BufferPool serves ONE step at a time."""
    h = torch.cuda.current_stream(device).cuda_stream
    idx = device.index if device.index is not None else torch.cuda.current_device()
    for (d, which), st in _side_streams.items():
        if d == idx and st.cuda_stream == h:
            return ("side", which)
    # The process-wide default pool has no owner that could promise "one step at a time": two eager callers on different user
    # streams must not share a zero-bordered operand buffer or a matching workspace, so there the handle is part of the key.
    if _pool is _default_pool:
        return ("main", h)
    return "main"


class fork:
    """with ops.fork(device, which) as f: ...   -- the body is enqueued on side stream `which`, which first waits for everything
    already on the current stream (and for the `after` events, if given); `f.join(*tensors)` afterwards makes the current stream
    wait for the side stream and hands the tensors made there over to it.  Works eagerly and inside hipGraph capture.

    Lifetime rule (explicit, no argument by allocator behaviour): every tensor that crosses a stream is `record_stream`ed on the
    stream that did not allocate it -- `f.use(t)` for a tensor of the current stream that the body reads, `f.join(t, ...)` for
    the tensors the body made -- so the caching allocator never hands its block out again before both streams are done with it."""

    def __init__(self, device, which=0, after=(), start=None):
        self.side = side_stream(device, which)
        self.cur = torch.cuda.current_stream(device)
        self.after = tuple(after) if isinstance(after, (tuple, list)) else (after,)
        self.start = start          # an event of the current stream: the body waits for IT instead of for everything enqueued so far
        self._ctx = None

    def __enter__(self):
        if self.start is not None:
            self.side.wait_event(self.start)
        else:
            self.side.wait_stream(self.cur)
        for ev in self.after:
            if ev is not None:
                self.side.wait_event(ev)           # waiting for an event of the same stream is free
        self._ctx = torch.cuda.stream(self.side)
        self._ctx.__enter__()
        return self

    def __exit__(self, *exc):
        self._ctx.__exit__(*exc)
        return False

    def use(self, *tensors):
        """Tensors allocated on the current stream that the forked body reads."""
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.side)

    def join(self, *tensors):
        """The current stream waits for the side stream; `tensors` (made on the side stream) may be used on it from here on."""
        self.cur.wait_stream(self.side)
        for t in tensors:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(self.cur)


def _dev(t, dtype, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor: the geoMatch ops have no CPU fallback" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


def _idx32(t, name):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError("%s must be a CUDA (HIP) tensor: the geoMatch ops have no CPU fallback" % name)
    if t.dtype == torch.int64:
        t = t.to(torch.int32)
    elif t.dtype != torch.int32:
        raise TypeError("%s must be int32 or int64, got %s" % (name, t.dtype))
    return t if t.is_contiguous() else t.contiguous()


# --------------------------------------------------------------------------------------
# neighbour search
# --------------------------------------------------------------------------------------
def knn_batch(support, query, K, return_d2=False):
    """support f32[B,S,3], query f32[B,Q,3] -> idx i32[B,Q,K] (ascending d2, ties by index)."""
    support = _dev(support, torch.float32, "support")
    query = _dev(query, torch.float32, "query")
    B, S, _ = support.shape
    Q = query.shape[1]
    assert support.shape[2] == 3 and query.shape[2] == 3 and query.shape[0] == B
    idx = torch.empty((B, Q, K), dtype=torch.int32, device=support.device)
    d2 = torch.empty((B, Q, K), dtype=torch.float32, device=support.device) if return_d2 else None
    check(_lib.lib().gdm_knn_batch_hip(support.data_ptr(), query.data_ptr(), B, S, Q, K, idx.data_ptr(),
                                       d2.data_ptr() if return_d2 else None, _stream()), "gdm_knn_batch_hip")
    return (idx, d2) if return_d2 else idx


def knn_jobs(jobs, B, keep_workspace=None):
    """jobs: list of (support f32 view [B,S,3], query f32 view [B,Q,3], K[, grid_w]).  Views may be prefix
    slices along dim 1 of contiguous [B,N,3] arrays (batch stride kept).  grid_w > 0 declares the support an organised map of
    S / grid_w rows x grid_w columns (row-major pixels of a depth crop): K > 1 searches then scan only the window of pixel
    columns / rows that can hold a neighbour (same results).  One launch per K class.  Returns the idx tensors i32[B,Q,K]."""
    n = len(jobs)
    arr = (KnnJob * n)()
    # ONE allocation for all index arrays (22 per pyramid): the host time of this function sits on the step's critical path (the
    # searches are its first kernels), and 22 allocator calls were a third of it
    sizes = []
    for job in jobs:
        sup, qry, K = job[:3]
        for t, nm in ((sup, "support"), (qry, "query")):
            if not t.is_cuda or t.dtype != torch.float32:
                raise RuntimeError("knn_jobs: %s must be a CUDA float32 tensor" % nm)
            if t.dim() != 3 or t.shape[0] != B or t.shape[2] != 3 or t.stride(2) != 1 or t.stride(1) != 3:
                raise ValueError("knn_jobs: %s must be [B,n,3] with unit point stride, got %s strides %s" %
                                 (nm, tuple(t.shape), t.stride()))
        sizes.append(B * qry.shape[1] * K)
    starts = []
    tot = 0
    for sz in sizes:                                          # every array starts on a 256-byte boundary (its readers use 16-byte loads)
        starts.append(tot)
        tot += (sz + 63) // 64 * 64
    slab = torch.empty(tot, dtype=torch.int32, device=jobs[0][0].device)
    base = slab.data_ptr()
    outs = []
    for i, job in enumerate(jobs):
        off = starts[i]
        sup, qry, K = job[:3]
        grid_w = int(job[3]) if len(job) > 3 else 0
        S, Q = sup.shape[1], qry.shape[1]
        a = arr[i]
        a.support = sup.data_ptr()
        a.query = qry.data_ptr()
        a.idx = base + 4 * off
        a.d2 = None
        a.support_bstride = sup.stride(0) if B > 1 else S * 3
        a.query_bstride = qry.stride(0) if B > 1 else Q * 3
        a.S, a.Q, a.K = S, Q, K
        a.grid_w = grid_w if (grid_w > 0 and S % grid_w == 0) else 0
        outs.append(slab[off:off + sizes[i]].view(B, Q, K))
    ws = _knn_launch(arr, n, B, jobs[0][0].device)
    if keep_workspace is not None:
        keep_workspace.append(ws)        # the caller keeps it alive as long as the results (explicit lifetime across streams)
    return outs


def copy_views(views):
    """Dense copies of a list of strided views in ONE launch (include/gdm.h gdm_copy_jobs_hip): every view is a 3-D or 4-D tensor
    of 4-byte elements whose last dimension is contiguous.  Returns the dense tensors (same shapes)."""
    n = len(views)
    arr = (_lib.CopyJob * n)()
    outs = []
    for i, v in enumerate(views):
        if not v.is_cuda or v.element_size() != 4 or v.dim() not in (3, 4) or v.stride(-1) != 1:
            raise ValueError("copy_views: view %d must be a 3-D / 4-D CUDA tensor of 4-byte elements with a contiguous last dimension" % i)
        out = torch.empty(v.shape, dtype=v.dtype, device=v.device)
        v4 = v if v.dim() == 4 else v.unsqueeze(1)
        arr[i] = _lib.CopyJob(out.data_ptr(), v.data_ptr(), v4.stride(0), v4.stride(1), v4.stride(2),
                              v4.shape[0], v4.shape[1], v4.shape[2], v4.shape[3])
        outs.append(out)
    check(_lib.lib().gdm_copy_jobs_hip(arr, n, _stream()), "gdm_copy_jobs_hip")
    return outs


def _knn_launch(arr, n, B, device):
    """gdm_knn_jobs_ws_hip with a workspace from torch's caching allocator (graph-capture safe): the K > 1 jobs' support sets are
    re-laid out once per launch as hashed float4 tiles."""
    L = _lib.lib()
    nbytes = int(L.gdm_knn_jobs_workspace_bytes(arr, n, B))
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=device)
    check(L.gdm_knn_jobs_ws_hip(arr, n, B, ws.data_ptr(), nbytes, _stream()), "gdm_knn_jobs_ws_hip")
    return ws


def ballquery(radius, nsample, xyz, new_xyz):
    """pointops.BallQuery.forward argument order (pointops.py:205): -> i32[B,m,nsample]."""
    xyz = _dev(xyz, torch.float32, "xyz")
    new_xyz = _dev(new_xyz, torch.float32, "new_xyz")
    B, n, _ = xyz.shape
    m = new_xyz.shape[1]
    idx = torch.zeros((B, m, nsample), dtype=torch.int32, device=xyz.device)
    check(_lib.lib().gdm_ballquery_hip(B, n, m, float(radius), nsample, new_xyz.data_ptr(), xyz.data_ptr(),
                                       idx.data_ptr(), _stream()), "gdm_ballquery_hip")
    return idx


def furthestsampling(xyz, m):
    """pointops.FurthestSampling.forward (pointops.py:40-50): xyz f32[B,n,3] -> i32[B,m]."""
    xyz = _dev(xyz, torch.float32, "xyz")
    B, n, _ = xyz.shape
    idx = torch.empty((B, m), dtype=torch.int32, device=xyz.device)
    temp = torch.empty((B, n), dtype=torch.float32, device=xyz.device)
    check(_lib.lib().gdm_furthestsampling_hip(B, n, m, xyz.data_ptr(), temp.data_ptr(), idx.data_ptr(), _stream()),
          "gdm_furthestsampling_hip")
    return idx


# --------------------------------------------------------------------------------------
# gather / pool, with backward
# --------------------------------------------------------------------------------------
def _feat3(feat):
    if feat.dim() == 4:
        assert feat.shape[3] == 1
        feat = feat.squeeze(3)
    return feat


class _GroupGather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        feat = _dev(feat, torch.float32, "feat")
        B, C, n = feat.shape
        m, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, C, m, K), dtype=torch.float32, device=feat.device)
        check(_lib.lib().gdm_group_gather_hip(feat.data_ptr(), idx.data_ptr(), B, C, n, m, K, out.data_ptr(), _stream()),
              "gdm_group_gather_hip")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, go):
        (idx,) = ctx.saved_tensors
        B, C, m, K = go.shape
        # a channel slice of a concatenation's gradient (dense rows, a larger batch stride) is read where it lies
        if not (go.dtype == torch.float32 and go.stride(3) == 1 and go.stride(2) == K and go.stride(1) == m * K and go.stride(0) >= C * m * K):
            go = go.contiguous()
        g = torch.zeros((B, C, ctx.n), dtype=torch.float32, device=go.device)
        check(_lib.lib().gdm_group_gather_bwd2_hip(go.data_ptr(), go.stride(0), idx.data_ptr(), B, C, ctx.n, m, K, g.data_ptr(), _stream()),
              "gdm_group_gather_bwd_hip")
        return g, None


def group_gather(feat, idx):
    """feat f32[B,C,n(,1)], idx int[B,m,K] -> f32[B,C,m,K]."""
    idx = _idx32(idx, "idx")
    assert idx.dim() == 3
    return _GroupGather.apply(_feat3(feat), idx)


def gather_nn(feat, idx):
    """feat f32[B,C,n(,1)], idx int[B,m] or [B,m,1] -> f32[B,C,m]."""
    idx = _idx32(idx, "idx")
    if idx.dim() == 2:
        idx = idx.unsqueeze(2)
    assert idx.shape[2] == 1
    return _GroupGather.apply(_feat3(feat), idx).squeeze(3)


class _GatherMax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, idx):
        feat = _dev(feat, torch.float32, "feat")
        B, C, n = feat.shape
        m, K = idx.shape[1], idx.shape[2]
        out = torch.empty((B, C, m), dtype=torch.float32, device=feat.device)
        need_arg = ctx.needs_input_grad[0]               # (not feat.requires_grad: _dev may have made a contiguous copy under no_grad)
        arg = torch.empty((B, C, m), dtype=torch.int32, device=feat.device) if need_arg else None
        check(_lib.lib().gdm_gather_max_hip(feat.data_ptr(), idx.data_ptr(), B, C, n, m, K, out.data_ptr(),
                                            arg.data_ptr() if need_arg else None, _stream()), "gdm_gather_max_hip")
        if need_arg:
            ctx.save_for_backward(arg)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, go):
        (arg,) = ctx.saved_tensors
        go = go.contiguous()
        B, C, m = go.shape
        g = torch.zeros((B, C, ctx.n), dtype=torch.float32, device=go.device)
        check(_lib.lib().gdm_gather_max_bwd_hip(go.data_ptr(), arg.data_ptr(), B, C, ctx.n, m, g.data_ptr(), _stream()),
              "gdm_gather_max_bwd_hip")
        return g, None


def gather_max(feat, idx):
    """feat f32[B,C,n(,1)], idx int[B,m,K] -> f32[B,C,m] = max over the K gathered neighbours."""
    idx = _idx32(idx, "idx")
    assert idx.dim() == 3
    return _GatherMax.apply(_feat3(feat), idx)


def rel_pos_enc(xyz, idx):
    """xyz f32[B,n,3], idx int[B,n,K] -> f32[B,10,n,K] (inputs are data: no gradient)."""
    xyz = _dev(xyz, torch.float32, "xyz")
    idx = _idx32(idx, "idx")
    B, n, _ = xyz.shape
    K = idx.shape[2]
    assert idx.shape[0] == B and idx.shape[1] == n
    out = torch.empty((B, 10, n, K), dtype=torch.float32, device=xyz.device)
    check(_lib.lib().gdm_rel_pos_enc_hip(xyz.data_ptr(), idx.data_ptr(), B, n, K, out.data_ptr(), _stream()),
          "gdm_rel_pos_enc_hip")
    return out


class _AttPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, att, feat):
        att = _dev(att, torch.float32, "att")
        feat = _dev(feat, torch.float32, "feat")
        B, C, n, K = att.shape
        assert feat.shape == att.shape
        out = torch.empty((B, C, n), dtype=torch.float32, device=att.device)
        check(_lib.lib().gdm_att_pool_hip(att.data_ptr(), feat.data_ptr(), B, C, n, K, out.data_ptr(), _stream()),
              "gdm_att_pool_hip")
        ctx.save_for_backward(att, feat)
        return out

    @staticmethod
    def backward(ctx, go):
        att, feat = ctx.saved_tensors
        go = go.contiguous()
        B, C, n, K = att.shape
        ga = torch.empty_like(att)
        gf = torch.empty_like(feat)
        check(_lib.lib().gdm_att_pool_bwd_hip(att.data_ptr(), feat.data_ptr(), go.data_ptr(), B, C, n, K,
                                              ga.data_ptr(), gf.data_ptr(), _stream()), "gdm_att_pool_bwd_hip")
        return ga, gf


def att_pool(att, feat):
    """att, feat f32[B,C,n,K] -> f32[B,C,n]: sum_k softmax_k(att) * feat."""
    return _AttPool.apply(att, feat)


def topk_rows(score, k, return_values=False):
    """score f32[..., n] -> idx i32[..., k] of the k largest per row (ties: lower column first)
    (dgcnn.py:26 `pairwise_distance.topk(k=k, dim=-1)[1]`)."""
    score = _dev(score, torch.float32, "score")
    n = score.shape[-1]
    rows = score.numel() // n
    idx = torch.empty(score.shape[:-1] + (k,), dtype=torch.int32, device=score.device)
    val = torch.empty(score.shape[:-1] + (k,), dtype=torch.float32, device=score.device) if return_values else None
    check(_lib.lib().gdm_topk_rows_hip(score.data_ptr(), rows, n, k, idx.data_ptr(), val.data_ptr() if return_values else None,
                                       _stream()), "gdm_topk_rows_hip")
    return (idx, val) if return_values else idx


def topk_negdist(gram, xx, k):
    """Indices of dgcnn.py:22-26 (`pairwise_distance.topk(k)`), from gram f32[B,n,n] = x^T x and xx f32[B,n] = sum_c x^2, without
    materialising pairwise_distance (same fp32 operations in the same order: identical indices)."""
    gram = _dev(gram, torch.float32, "gram")
    xx = _dev(xx, torch.float32, "xx")
    B, n, _ = gram.shape
    idx = torch.empty((B, n, k), dtype=torch.int32, device=gram.device)
    check(_lib.lib().gdm_topk_negdist_hip(gram.data_ptr(), xx.data_ptr(), B, n, k, idx.data_ptr(), _stream()), "gdm_topk_negdist_hip")
    return idx


def affine_act_maxk(x, scale, shift, act=0, slope=0.0):
    """max over the last (neighbour) dimension of act(scale[c]*x + shift[c]); x f32[B,C,n,K] -> f32[B,C,n].  Inference only."""
    x = _dev(x, torch.float32, "x")
    B, C, n, K = x.shape
    out = torch.empty((B, C, n), dtype=torch.float32, device=x.device)
    check(_lib.lib().gdm_affine_act_maxk_hip(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), B * C, C, n, K, act, float(slope),
                                             out.data_ptr(), _stream()), "gdm_affine_act_maxk_hip")
    return out


class _EdgeFeature(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, idx):
        x = _dev(x, torch.float32, "x")
        B, C, n = x.shape
        K = idx.shape[2]
        out = torch.empty((B, 2 * C, n, K), dtype=torch.float32, device=x.device)
        check(_lib.lib().gdm_edge_feature_hip(x.data_ptr(), idx.data_ptr(), B, C, n, K, out.data_ptr(), _stream()),
              "gdm_edge_feature_hip")
        ctx.save_for_backward(idx)
        return out

    @staticmethod
    def backward(ctx, go):
        (idx,) = ctx.saved_tensors
        go = go.contiguous()
        B, C2, n, K = go.shape
        g = torch.zeros((B, C2 // 2, n), dtype=torch.float32, device=go.device)
        check(_lib.lib().gdm_edge_feature_bwd_hip(go.data_ptr(), idx.data_ptr(), B, C2 // 2, n, K, g.data_ptr(), _stream()),
              "gdm_edge_feature_bwd_hip")
        return g, None


def edge_feature(x, idx):
    """x f32[B,C,n], idx int[B,n,K] -> f32[B,2C,n,K] = cat(x_j - x_i, x_i) (dgcnn.py:30-56 get_graph_feature)."""
    idx = _idx32(idx, "idx")
    return _EdgeFeature.apply(x, idx)


class _CircleRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sim, match, item, xyz, vis, radius, gamma, m):
        sim = _dev(sim, torch.float32, "sim")
        R, Mp = sim.shape
        lse_p = torch.empty(R, dtype=torch.float32, device=sim.device)
        lse_n = torch.empty_like(lse_p)
        loss = torch.empty_like(lse_p)
        check(_lib.lib().gdm_circle_rows_fwd_hip(sim.data_ptr(), R, Mp, match.data_ptr(), item.data_ptr(), xyz.data_ptr(),
                                                 vis.data_ptr(), radius, gamma, m, lse_p.data_ptr(), lse_n.data_ptr(),
                                                 loss.data_ptr(), _stream()), "gdm_circle_rows_fwd_hip")
        ctx.save_for_backward(sim, match, item, xyz, vis, lse_p, lse_n)
        ctx.consts = (radius, gamma, m)
        return loss

    @staticmethod
    def backward(ctx, g):
        sim, match, item, xyz, vis, lse_p, lse_n = ctx.saved_tensors
        radius, gamma, m = ctx.consts
        g = g.contiguous()
        R, Mp = sim.shape
        dsim = torch.empty_like(sim)
        check(_lib.lib().gdm_circle_rows_bwd_hip(sim.data_ptr(), R, Mp, match.data_ptr(), item.data_ptr(), xyz.data_ptr(),
                                                 vis.data_ptr(), radius, gamma, m, lse_p.data_ptr(), lse_n.data_ptr(),
                                                 g.data_ptr(), dsim.data_ptr(), _stream()), "gdm_circle_rows_bwd_hip")
        return dsim, None, None, None, None, None, None, None


def circle_rows(sim, match, item, xyz, vis, radius, gamma=16.0, m=0.2):
    """Per-row circle loss with the on-the-fly positive mask (geoMatch.py:55-83 + loss.py:470-494).
    sim f32[R,M+1], match int[R] (M = none), item int[R], xyz f32[M,3], vis (any dtype, nonzero = visible)[B,M] -> f32[R]."""
    match = _idx32(match, "match")
    item = _idx32(item, "item")
    xyz = _dev(xyz, torch.float32, "xyz")
    if not vis.is_cuda:
        raise RuntimeError("vis must be a CUDA (HIP) tensor: the geoMatch ops have no CPU fallback")
    vis8 = (vis != 0).to(torch.uint8).contiguous()
    return _CircleRows.apply(sim, match, item, xyz, vis8, float(radius), float(gamma), float(m))


# --------------------------------------------------------------------------------------
# training matching loss without the similarity matrix (gdm_circle.hip)
# --------------------------------------------------------------------------------------
def circle_nbr_table(xyz, radius):
    """Bit table u32[M, ceil(M/32)]: vertices within `radius` of each vertex (the reference's pdist arithmetic).  Depends on the
    model only: build once per (xyz, radius) and reuse."""
    xyz = _dev(xyz, torch.float32, "xyz")
    M = xyz.shape[0]
    nbr = torch.empty((M, (M + 31) // 32), dtype=torch.int32, device=xyz.device)
    check(_lib.lib().gdm_circle_match_nbr_hip(xyz.data_ptr(), M, float(radius), nbr.data_ptr(), _stream()), "gdm_circle_match_nbr_hip")
    return nbr


def circle_nbr_items(xyz, rad):
    """Per-item positive tables u32[B, M, ceil(M/32)] for a per-item, per-vertex radius rad f32[B,M]: bit c of row (b, g) =
    sqrt(|xyz_g - xyz_c|^2 + 1e-7) < rad[b, c] (the geoMatch_DGCNN variant, geoMatch_DGCNN.py:62-70)."""
    xyz = _dev(xyz, torch.float32, "xyz")
    rad = _dev(rad, torch.float32, "rad")
    B, M = rad.shape
    assert xyz.shape == (M, 3)
    nbr = torch.empty((B, M, (M + 31) // 32), dtype=torch.int32, device=xyz.device)
    check(_lib.lib().gdm_circle_match_nbr_items_hip(xyz.data_ptr(), M, rad.data_ptr(), B, nbr.data_ptr(), _stream()),
          "gdm_circle_match_nbr_items_hip")
    return nbr


def circle_visbits(vis):
    """visible_flag (any dtype, nonzero = visible) [B,M] -> bits i32[B, ceil(M/32)]."""
    if not vis.is_cuda:
        raise RuntimeError("vis must be a CUDA (HIP) tensor: the geoMatch ops have no CPU fallback")
    v8 = (vis != 0).to(torch.uint8).contiguous()
    B, M = v8.shape
    bits = torch.empty((B, (M + 31) // 32), dtype=torch.int32, device=vis.device)
    check(_lib.lib().gdm_circle_match_visbits_hip(v8.data_ptr(), B, M, bits.data_ptr(), _stream()), "gdm_circle_match_visbits_hip")
    return bits


def _cm_pack(x):
    L = _lib.lib()
    n = x.shape[0]
    npad = (n + 127) // 128 * 128
    rows = torch.empty(L.gdm_circle_match_rows_bytes(n), dtype=torch.uint8, device=x.device)
    tp = torch.empty(L.gdm_circle_match_tp_bytes(n), dtype=torch.uint8, device=x.device)
    rsum = torch.empty(npad, dtype=torch.float32, device=x.device)
    check(L.gdm_circle_match_pack_hip(x.data_ptr(), n, rows.data_ptr(), tp.data_ptr(), rsum.data_ptr(), _stream()), "gdm_circle_match_pack_hip")
    return rows, tp, rsum


def _pad_rows(t, npad, fill):
    if t.shape[0] == npad:
        return t.contiguous()
    out = torch.full((npad,) + tuple(t.shape[1:]), fill, dtype=t.dtype, device=t.device)
    out[: t.shape[0]] = t
    return out


class _CircleMatch(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, g, c2, item, nbr, visb, gamma, m, pad_e0=False):
        x = _dev(x, torch.float32, "x")
        y = _dev(y, torch.float32, "y")
        R, M = x.shape[0], y.shape[0]
        assert x.shape[1] == 128 and y.shape[1] == 128, "descriptors must have 128 channels"
        Rp = (R + 127) // 128 * 128
        xr, xt, xs = _cm_pack(x)
        yr, yt, _ = _cm_pack(y)
        per_item = nbr is not None and nbr.dim() == 3            # [B, M, W]: one table per batch item (circle_nbr_items)
        if pad_e0:
            xs = _pad_rows(x[:, 0].contiguous(), Rp, 0.0)          # <x_r, e0>: the similarity with the padding column
        gp = _pad_rows(g, Rp, M)
        ip = _pad_rows(item, Rp, 0)
        c2p = _pad_rows(c2, Rp, -1) if c2 is not None else None
        lp = torch.empty(Rp, dtype=torch.float32, device=x.device)
        ln = torch.empty_like(lp)
        loss = torch.empty_like(lp)
        check(_lib.lib().gdm_circle_match_fwd2_hip(xr.data_ptr(), xt.data_ptr(), xs.data_ptr(), yr.data_ptr(), yt.data_ptr(), R, M,
                                                   gp.data_ptr(), c2p.data_ptr() if c2p is not None else None, ip.data_ptr(),
                                                   nbr.data_ptr() if nbr is not None else None, int(per_item),
                                                   visb.data_ptr() if visb is not None else None, int(bool(pad_e0)),
                                                   gamma, m, lp.data_ptr(), ln.data_ptr(), loss.data_ptr(), _stream()), "gdm_circle_match_fwd_hip")
        ctx.save_for_backward(xr, xt, xs, yr, yt, gp, ip, lp, ln)
        ctx.extra = (c2p, nbr, visb, gamma, m, R, M, per_item, bool(pad_e0))
        return loss[:R]

    @staticmethod
    def backward(ctx, gout):
        xr, xt, xs, yr, yt, gp, ip, lp, ln = ctx.saved_tensors
        c2p, nbr, visb, gamma, m, R, M, per_item, pad_e0 = ctx.extra
        L = _lib.lib()
        Rp, Mp = lp.shape[0], (M + 127) // 128 * 128
        z = lp[:R] + ln[:R]
        sig = torch.where(z > 20.0, torch.ones_like(z), torch.sigmoid(z))            # d softplus
        coef = torch.where(torch.isfinite(lp[:R]), gout.contiguous().float() * sig, torch.zeros_like(z))   # empty positive set: 0
        coef = _pad_rows(coef, Rp, 0.0)
        P = int(L.gdm_circle_match_bwd_parts(R, M))
        gx = torch.empty((Rp, 128), dtype=torch.float32, device=lp.device)
        gyp = torch.empty((P, Mp, 128), dtype=torch.float32, device=lp.device)
        check(L.gdm_circle_match_bwd2_hip(xr.data_ptr(), xt.data_ptr(), xs.data_ptr(), yr.data_ptr(), yt.data_ptr(), R, M, gp.data_ptr(),
                                          c2p.data_ptr() if c2p is not None else None, ip.data_ptr(),
                                          nbr.data_ptr() if nbr is not None else None, int(per_item),
                                          visb.data_ptr() if visb is not None else None, int(pad_e0),
                                          gamma, m, lp.data_ptr(), ln.data_ptr(), coef.data_ptr(), gx.data_ptr(), gyp.data_ptr(), _stream()),
              "gdm_circle_match_bwd_hip")
        return gx[:R], gyp.sum(dim=0)[:M], None, None, None, None, None, None, None, None


def circle_match(x, y, g, item, nbr=None, visb=None, c2=None, gamma=16.0, m=0.2, pad="ones"):
    """Per-row circle loss of unit scene rows x f32[R,128] against unit vertex rows y f32[M,128] WITHOUT the [R, M+1] similarity
    matrix (geoMatch.py:117-136 + :55-100 + loss.py:441-494, forward and backward).  g int[R]: ground-truth vertex (M = none),
    item int[R]: batch item; positives from `nbr` (circle_nbr_table) & `visb` (circle_visbits), or -- symmetric objects -- the two
    columns g / c2 of each row.  `nbr` may be one table per batch item (circle_nbr_items, [B, M, W]); pad = "ones": the padding column
    is -1/sqrt(128) everywhere (geoMatch.py:117-119), "e0": the unit vector e0 (geoMatch_DGCNN.py:96-99).  Returns f32[R];
    differentiable w.r.t. x and y."""
    if pad not in ("ones", "e0"):
        raise ValueError("circle_match: pad=%r" % (pad,))
    g = _idx32(g, "g")
    item = _idx32(item, "item")
    if c2 is not None:
        c2 = _idx32(c2, "c2")
    elif nbr is None or visb is None:
        raise ValueError("circle_match: pass nbr and visb (radius-test positives) or c2 (symmetric objects)")
    return _CircleMatch.apply(x, y, g, c2, item, nbr, visb, float(gamma), float(m), pad == "e0")


ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def affine_act(x, scale, shift, act=ACT_NONE, slope=0.0, res=None, res_scale=None, res_shift=None, inplace=True):
    """Inference only (no autograd): y = act(x*scale[c] + shift[c] (+ res*res_scale[c] + res_shift[c])) for
    x f32[B,C,...] contiguous.  Folded eval BatchNorm + activation (+ residual) in one launch."""
    x = _dev(x, torch.float32, "x")
    B, C = x.shape[0], x.shape[1]
    inner = x.numel() // (B * C)
    if inner % 4 != 0 or B * C > 65535 or x.data_ptr() % 16 != 0:
        y = x * scale.view(1, C, *([1] * (x.dim() - 2))) + shift.view(1, C, *([1] * (x.dim() - 2)))
        if res is not None:
            y = y + (res * res_scale.view_as(scale.view(1, C, *([1] * (x.dim() - 2)))) + res_shift.view(1, C, *([1] * (x.dim() - 2)))
                     if res_scale is not None else res)
        if act == ACT_RELU:
            y = torch.relu(y)
        elif act == ACT_LEAKY:
            y = torch.where(y > 0, y, y * slope)
        return y
    if res is not None:
        res = _dev(res, torch.float32, "res")
        assert res.shape == x.shape
    y = x if inplace else torch.empty_like(x)
    check(_lib.lib().gdm_affine_act_hip(x.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                        res.data_ptr() if res is not None else None,
                                        res_scale.data_ptr() if res_scale is not None else None,
                                        res_shift.data_ptr() if res_shift is not None else None,
                                        B * C, C, inner, act, float(slope), y.data_ptr(), _stream()), "gdm_affine_act_hip")
    return y


def _pw_seg(spec, B, name):
    """(x f32[B,C,n_src(,1)] channel-major[, idx int[B,n(,1)]]) -> (_lib.PwSeg, tensors to keep alive, n or None)."""
    x, idx = (spec, None) if torch.is_tensor(spec) else (spec[0], spec[1])
    x = _dev(x, torch.float32, name)
    if x.dim() == 4 and x.shape[3] == 1:
        x = x.reshape(x.shape[0], x.shape[1], x.shape[2])
    if x.dim() != 3 or x.shape[0] != B:
        raise ValueError("pointwise: %s must be [B,C,n(,1)] with B=%d, got %s" % (name, B, tuple(x.shape)))
    n = None
    if idx is not None:
        idx = _idx32(idx, name + " index").reshape(B, -1)
        n = idx.shape[1]
    seg = _lib.PwSeg(x.data_ptr(), idx.data_ptr() if idx is not None else None, x.shape[1], x.shape[2])
    return seg, (x, idx), n


def pointwise(segs, wt, scale=None, shift=None, act=ACT_NONE, slope=0.0, point_major=False, out=None, out_c0=0, w_rowmajor=False):
    """One per-point (1x1) layer in one launch (include/gdm.h gdm_pointwise_hip), inference only:
        y[b,:,i] = act(scale * (W . cat(segs)[b,:,i]) + shift)
    segs: a list of one to three of  x f32[B,C,n(,1)]  or  (x f32[B,C,n_src(,1)], idx int[B,n(,1)])  -- the concat along channels
    is never formed, an indexed segment is read through its index (nearest-neighbour interpolation folded into the load).
    wt f32[K,Cout]: the layer's weight TRANSPOSED (K = total input channels); w_rowmajor: wt is f32[Cout,K] instead, the weight as an
    nn.Conv1d / nn.Conv2d holds it (the training path: no transposed copy of a weight that changes every step).
    Returns f32[B,Cout,n], or f32[B,n,Cout] when point_major; with `out` (f32[B,outC,n] / [B,n,outC]) channels
    [out_c0, out_c0+Cout) of it are written instead."""
    if not isinstance(segs, list):
        segs = [segs]
    first = segs[0] if torch.is_tensor(segs[0]) else segs[0][0]
    B = first.shape[0]
    wt = _dev(wt, torch.float32, "wt")
    K, Cout = (wt.shape[1], wt.shape[0]) if w_rowmajor else wt.shape
    if w_rowmajor and K >= 32 and B * first.numel() // max(1, first.shape[0] * first.shape[1]) >= 131072:
        # many points: the kernel's weight fragment loads run along Cout ([K, Cout]: one 64-byte piece per K row; from [Cout, K] sixteen
        # cache lines per load -- 76 vs 52 us at 64 -> 64 channels, 393 k points), so the few-KB transposed copy pays for itself
        wt, w_rowmajor = wt.t().contiguous(), False
    arr = (_lib.PwSeg * len(segs))()
    keep = []
    n = None
    ksum = 0
    for i, sp in enumerate(segs):
        seg, k, ni = _pw_seg(sp, B, "segment %d" % i)
        arr[i] = seg
        keep.append(k)
        ksum += seg.C
        ni = ni if ni is not None else seg.n_src
        if n is not None and ni != n:
            raise ValueError("pointwise: segments disagree on the number of points (%d vs %d)" % (n, ni))
        n = ni
    if ksum != K:
        raise ValueError("pointwise: weight has %d input rows, the segments %d channels" % (K, ksum))
    if out is None:
        out = torch.empty((B, n, Cout) if point_major else (B, Cout, n), dtype=torch.float32, device=wt.device)
        outC = Cout
    else:
        outC = out.shape[2] if point_major else out.shape[1]
        if not out.is_contiguous() or out.dtype != torch.float32 or out.shape[0] != B or (out.shape[1] if point_major else out.shape[2]) != n:
            raise ValueError("pointwise: out must be a contiguous f32 [B,%s] tensor" % ("n,outC" if point_major else "outC,n"))
    check(_lib.lib().gdm_pointwise2_hip(arr, len(segs), wt.data_ptr(), 1 if w_rowmajor else 0, scale.data_ptr() if scale is not None else None,
                                        shift.data_ptr() if shift is not None else None, B, n, Cout, int(act), float(slope),
                                        out.data_ptr(), outC, int(out_c0), 1 if point_major else 0, _stream()), "gdm_pointwise_hip")
    return out


def pointwise_chain2(x, p0, p1):
    """Two chained narrow per-point layers in one launch (include/gdm.h gdm_pointwise_chain2_hip): x f32[B,C0,n]; p0, p1 =
    (wt [K,Cout], scale, shift, act, slope) as `_FusedConvMixin._pointwise_params` gives them -> (y0 [B,C1,n], y1 [B,C2,n])."""
    x = _dev(x, torch.float32, "x")
    B, C0, n = x.shape
    (w0, s0, b0, a0, sl0), (w1, s1, b1, a1, sl1) = p0, p1
    C1, C2 = w0.shape[1], w1.shape[1]
    if w0.shape[0] != C0 or w1.shape[0] != C1:
        raise ValueError("pointwise_chain2: weights %s, %s do not chain from %d channels" % (tuple(w0.shape), tuple(w1.shape), C0))
    y0 = torch.empty((B, C1, n), dtype=torch.float32, device=x.device)
    y1 = torch.empty((B, C2, n), dtype=torch.float32, device=x.device)
    ptr = lambda t: t.data_ptr() if t is not None else None
    check(_lib.lib().gdm_pointwise_chain2_hip(x.data_ptr(), w0.data_ptr(), ptr(s0), ptr(b0), int(a0), float(sl0), w1.data_ptr(), ptr(s1), ptr(b1),
                                              int(a1), float(sl1), B, n, C0, C1, C2, y0.data_ptr(), y1.data_ptr(), _stream()),
          "gdm_pointwise_chain2_hip")
    return y0, y1


def pointwise_jobs(xs, wts):
    """Up to four independent plain per-point layers of equal K and Cout in ONE launch (include/gdm.h gdm_pointwise_jobs_hip):
    xs[j] f32[B,K,n_j], wts[j] f32[K,Cout] (the weight transposed) -> [f32[B,Cout,n_j]].  Bit-identical to `pointwise([x], wt)` per
    job (same tile function, same K split); jobs whose K splits would differ alone go in one launch per distinct split."""
    if not (1 <= len(xs) <= 4 and len(xs) == len(wts)):
        raise ValueError("pointwise_jobs: one to four (x, wt) pairs")
    B, K = xs[0].shape[0], xs[0].shape[1]
    Cout = wts[0].shape[1]
    arr = (_lib.PwJob * len(xs))()
    keep, outs = [], []
    for j, (x, wt) in enumerate(zip(xs, wts)):
        x = _dev(x.reshape(x.shape[0], x.shape[1], -1), torch.float32, "xs[%d]" % j)
        wt = _dev(wt, torch.float32, "wts[%d]" % j)
        if x.shape[0] != B or x.shape[1] != K or tuple(wt.shape) != (K, Cout):
            raise ValueError("pointwise_jobs: job %d has x %s / wt %s, job 0 has B=%d K=%d Cout=%d" % (j, tuple(x.shape), tuple(wt.shape), B, K, Cout))
        out = torch.empty((B, Cout, x.shape[2]), dtype=torch.float32, device=x.device)
        arr[j] = _lib.PwJob(x.data_ptr(), wt.data_ptr(), out.data_ptr(), x.shape[2])
        keep += [x, wt]
        outs.append(out)
    check(_lib.lib().gdm_pointwise_jobs_hip(arr, len(xs), B, K, Cout, _stream()), "gdm_pointwise_jobs_hip")
    return outs


def _fold_and_all_reduce(sums, C, group):
    """Partial pairs [G][C][2] + count + G  ->  double[2C+1] = one pair per channel + count, summed over the ranks of `group`."""
    import torch.distributed as dist
    compact = torch.cat([sums[:-2].view(-1, 2 * C).sum(0), sums[-2:-1]])
    if dist.get_backend(group) == "nccl":
        dist.all_reduce(compact, group=group)
    else:                                                   # gloo rehearsals: through host memory
        host = compact.cpu()
        dist.all_reduce(host, group=group)
        compact = host.to(sums.device)
    return compact


class _BatchNormAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, eps, momentum, act, slope, group):
        x = _dev(x, torch.float32, "x")
        B, C = x.shape[0], x.shape[1]
        inner = x.numel() // (B * C)
        L = _lib.lib()
        sums = torch.empty(L.gdm_bn_sums_len(B, C, inner), dtype=torch.float64, device=x.device)
        saved = torch.empty(4 * C, dtype=torch.float32, device=x.device)
        y = torch.empty_like(x)
        check(L.gdm_bn_stats_hip(x.data_ptr(), B, C, inner, sums.data_ptr(), _stream()), "gdm_bn_stats_hip")
        groups = 0
        if group is not None:                               # SyncBatchNorm: statistics over the whole data-parallel batch
            sums, groups = _fold_and_all_reduce(sums, C, group), 1
        check(L.gdm_bn_fwd_apply_hip(x.data_ptr(), sums.data_ptr(), groups, weight.data_ptr(), bias.data_ptr(), B, C, inner, float(eps),
                                     float(momentum), act, float(slope), saved.data_ptr(),
                                     running_mean.data_ptr() if running_mean is not None else None,
                                     running_var.data_ptr() if running_var is not None else None, y.data_ptr(), _stream()), "gdm_bn_fwd_apply_hip")
        ctx.save_for_backward(x, weight, saved)
        ctx.act, ctx.slope, ctx.group = act, float(slope), group
        return y

    @staticmethod
    def backward(ctx, go):
        x, weight, saved = ctx.saved_tensors
        go = _dev(go, torch.float32, "grad")
        B, C = x.shape[0], x.shape[1]
        inner = x.numel() // (B * C)
        L = _lib.lib()
        sums = torch.empty(L.gdm_bn_sums_len(B, C, inner), dtype=torch.float64, device=x.device)
        gw = torch.empty(C, dtype=torch.float32, device=x.device)
        gb = torch.empty(C, dtype=torch.float32, device=x.device)
        gx = torch.empty_like(x)
        check(L.gdm_bn_bwd_reduce_hip(x.data_ptr(), go.data_ptr(), saved.data_ptr(), B, C, inner, ctx.act, ctx.slope, sums.data_ptr(), _stream()),
              "gdm_bn_bwd_reduce_hip")
        groups = 0
        local = None
        if ctx.group is not None:
            # grad_x needs the sums over the whole batch; grad weight / bias stay LOCAL sums (DDP averages parameter gradients over
            # the ranks afterwards, as with nn.SyncBatchNorm)
            local = sums[:-2].view(-1, C, 2).sum(0)
            sums, groups = _fold_and_all_reduce(sums, C, ctx.group), 1
        check(L.gdm_bn_bwd_apply_hip(x.data_ptr(), go.data_ptr(), sums.data_ptr(), groups, weight.data_ptr(), saved.data_ptr(), B, C, inner,
                                     ctx.act, ctx.slope, gw.data_ptr(), gb.data_ptr(), gx.data_ptr(), _stream()), "gdm_bn_bwd_apply_hip")
        if local is not None:
            mean, rstd = saved[2 * C:3 * C].double(), saved[3 * C:].double()
            gb = local[:, 0].float()
            gw = (rstd * (local[:, 1] - mean * local[:, 0])).float()
        return gx, gw, gb, None, None, None, None, None, None, None




def _sync_group(bn):
    """Process group of a SyncBatchNorm that actually spans several ranks, else None."""
    import torch.distributed as dist
    if not (isinstance(bn, torch.nn.SyncBatchNorm) and dist.is_available() and dist.is_initialized()):
        return None
    group = bn.process_group if bn.process_group is not None else dist.group.WORLD
    return group if dist.get_world_size(group) >= settings.SYNCBN_MIN_WORLD else None


def bn_train_supported(x, bn):
    """Affine BatchNorm{1,2}d -- or SyncBatchNorm -- in training mode with torch-style running statistics, on a contiguous f32 GPU map
    whose inner size is a multiple of 4."""
    if not (settings.USE_FUSED_BN_TRAIN and bn.training and x.is_cuda and x.dtype == torch.float32 and x.dim() >= 3 and x.is_contiguous()):
        return False
    if type(bn) not in (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d) and not (settings.USE_FUSED_SYNCBN and type(bn) is torch.nn.SyncBatchNorm):
        return False
    if not bn.affine or bn.momentum is None or not bn.track_running_stats:
        return False
    B, C = x.shape[0], x.shape[1]
    inner = x.numel() // max(B * C, 1)
    return B * C <= 65535 and inner >= 4 and inner % 4 == 0 and x.data_ptr() % 16 == 0 and B * inner > 1


def batch_norm_act_train(x, bn, act=ACT_NONE, slope=0.0):
    """Training-mode `act(bn(x))` (batch statistics -- over all ranks for a SyncBatchNorm -- and running statistics updated as the
    module does) with a fused backward.  act: ACT_NONE / ACT_RELU / ACT_LEAKY(slope).  Caller checks bn_train_supported."""
    y = _BatchNormAct.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum, act, slope, _sync_group(bn))
    if bn.num_batches_tracked is not None:
        if _bn_counters is not None:
            _bn_counters.append(bn.num_batches_tracked)           # one multi-tensor add for the whole forward (batched_bn_counters)
        else:
            bn.num_batches_tracked.add_(1)
    return y


_bn_counters = None


class batched_bn_counters:
    """with ops.batched_bn_counters(): ... -- the `num_batches_tracked += 1` of every fused training BatchNorm inside the scope is applied
    at its exit as ONE multi-tensor add instead of one tiny launch per layer (91 launches, 0.34 ms of a training step).  Same values
    after the scope as the modules leave; nothing reads the counters inside a forward (momentum is a number in every layer here)."""

    def __enter__(self):
        global _bn_counters
        self.prev, _bn_counters = _bn_counters, []
        return self

    def __exit__(self, *exc):
        global _bn_counters
        pending, _bn_counters = _bn_counters, self.prev
        if pending:
            torch._foreach_add_(pending, 1)
        return False


def upconv3x3_gather(z, scale, shift, cout, out_size, act=ACT_NONE, slope=0.0, packed=False):
    """z f32[B, 9*cout, H, W] (low-resolution tap-major channel mixes) -> f32[B, cout, OH, OW]: the 9-tap bilinear
    gather that completes conv3x3(upsample(x)) + folded BN + activation.  Inference only.  packed: the eight-channel form, which also
    writes the packed split-bf16 operand of the GEMM that reads the map next (hung on the result as `_gdm_packed`)."""
    z = _dev(z, torch.float32, "z")
    B, c9, H, W = z.shape
    assert c9 == 9 * cout
    OH, OW = int(out_size[0]), int(out_size[1])
    out = torch.empty((B, cout, OH, OW), dtype=torch.float32, device=z.device)
    if (packed and (cout == 64 or cout % 128 == 0) and OW % 32 == 0 and (B * OH * OW) % 256 == 0 and B * cout // 8 <= 65535
            and (OH, OW) == (2 * H, 2 * W)):
        opk = PackedAct(_packed_buffer(B, cout, OH, OW, z.device), (B, cout, OH, OW))
        check(_lib.lib().gdm_upconv3x3_gather2_hip(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, cout, H, W, OH, OW, act,
                                                   float(slope), out.data_ptr(), opk.buf.data_ptr(), _stream()), "gdm_upconv3x3_gather2_hip")
        out._gdm_packed = opk
        return out
    check(_lib.lib().gdm_upconv3x3_gather_hip(z.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, cout, H, W, OH, OW, act,
                                              float(slope), out.data_ptr(), _stream()), "gdm_upconv3x3_gather_hip")
    return out


def upconv_fused64_pack_weight(weight):
    """conv weight f32[64,64,3,3] -> packed split-bf16 rows for upconv_fused64."""
    w = _dev(weight.detach(), torch.float32, "weight")
    if tuple(w.shape) != (64, 64, 3, 3):
        raise ValueError("upconv_fused64_pack_weight: weight must be [64,64,3,3], got %s" % (tuple(w.shape),))
    L = _lib.lib()
    wpk = torch.empty(L.gdm_upconv_fused64_weight_bytes(), dtype=torch.uint8, device=w.device)
    check(L.gdm_upconv_fused64_pack_weight_hip(w.data_ptr(), wpk.data_ptr(), _stream()), "gdm_upconv_fused64_pack_weight_hip")
    return wpk


def upconv_fused64(x, wpk, scale, shift, out_size, act=0, slope=0.0):
    """act(scale * conv3x3(upsample_bilinear_ac(x)) + shift) for 64 -> 64 channels in one kernel.  Inference only."""
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    OH, OW = int(out_size[0]), int(out_size[1])
    out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
    check(_lib.lib().gdm_upconv_fused64_hip(x.data_ptr(), wpk.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, C, H, W, OH, OW, act,
                                            float(slope), out.data_ptr(), _stream()), "gdm_upconv_fused64_hip")
    return out


class _UpconvGather(torch.autograd.Function):
    """out = 9-tap bilinear gather of z (+ bias[co]); differentiable in z and bias (training form of PSPUpsample)."""

    @staticmethod
    def forward(ctx, z, bias, cout, OH, OW):
        z = z.contiguous()
        B, c9, H, W = z.shape
        ones = torch.ones(cout, dtype=torch.float32, device=z.device)
        shift = bias.detach().contiguous() if bias is not None else torch.zeros(cout, dtype=torch.float32, device=z.device)
        out = torch.empty((B, cout, OH, OW), dtype=torch.float32, device=z.device)
        check(_lib.lib().gdm_upconv3x3_gather_hip(z.data_ptr(), ones.data_ptr(), shift.data_ptr(), B, cout, H, W, OH, OW, ACT_NONE, 0.0,
                                                  out.data_ptr(), _stream()), "gdm_upconv3x3_gather_hip")
        ctx.shape = (B, cout, H, W, OH, OW)
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, go):
        B, cout, H, W, OH, OW = ctx.shape
        go = go.contiguous()
        gz = torch.empty((B, 9 * cout, H, W), dtype=torch.float32, device=go.device)
        check(_lib.lib().gdm_upconv3x3_gather_bwd_hip(go.data_ptr(), B, cout, H, W, OH, OW, gz.data_ptr(), _stream()),
              "gdm_upconv3x3_gather_bwd_hip")
        return gz, (channel_sum(go) if ctx.has_bias else None), None, None, None


def upconv3x3_gather_train(z, bias, cout, out_size):
    """Differentiable form of upconv3x3_gather without the folded BN / activation: z f32[B,9*cout,H,W] -> f32[B,cout,OH,OW] (+ bias)."""
    z = _dev(z, torch.float32, "z")
    assert z.shape[1] == 9 * cout
    return _UpconvGather.apply(z, bias, cout, int(out_size[0]), int(out_size[1]))


def wx(w2d, x3):
    """W f32[Cout,Cin] . x f32[B,Cin,n] -> f32[B,Cout,n] as ONE batched GEMM on the shared weight.  torch.matmul(2-D, 3-D) folds the
    batch into GEMM rows instead and pays a transposing copy of the product (and of its gradient) to get back to [B,Cout,n]:
    2 ms of an 83 ms training step on the 9*Cout-channel tap tensors of PSPUpsample."""
    if torch.is_grad_enabled() and (w2d.requires_grad or x3.requires_grad) and x3.is_cuda and x3.dtype == torch.float32:
        return _WxTrain.apply(x3.contiguous(), w2d.contiguous())
    return torch.bmm(w2d.unsqueeze(0).expand(x3.shape[0], -1, -1), x3)


def _wgrad_parts(nchunk, tiles):
    """K split of a pixel-contraction GEMM: the largest divisor of the chunk count that keeps the launch near one round of
    workgroups (a part has `tiles` of them) and at least four chunks per part."""
    parts = 1
    for d in range(1, nchunk + 1):
        if nchunk % d == 0 and d * tiles <= 320 and nchunk // d >= 4:
            parts = d
    return parts


def gemm_wgrad_supported(x3, go3):
    B, Cin, P = x3.shape
    return (settings.USE_MFMA_GEMM_TRAIN and x3.is_cuda and x3.dtype == torch.float32 and P % 128 == 0 and Cin % 256 == 0
            and go3.shape[1] >= 64 and 2.0 * B * P * Cin * go3.shape[1] >= 2e9)


def gemm_wgrad(x3, go3):
    """dW f32[Cout,Cin] = sum_b go3[b] . x3[b]^T (x3 f32[B,Cin,P], go3 f32[B,Cout,P]) on the split-bf16 MFMA GEMM with the pixels as the
    contraction axis (include/gdm.h gdm_wgrad_pack_x1_hip): the weight gradient of a 1x1 convolution / of `wx`."""
    x3 = _dev(x3, torch.float32, "x")
    go3 = _dev(go3, torch.float32, "grad_out")
    B, Cin, P = x3.shape
    Cout = go3.shape[1]
    L = _lib.lib()
    nx, ng = L.gdm_wgrad_x1_bytes(B, Cin, P), L.gdm_wgrad_go_bytes(B, Cout, P // 32, 32)
    if nx == 0 or ng == 0 or Cin % 256 != 0:
        raise ValueError("gemm_wgrad: unsupported shape x %s grad_out %s" % (tuple(x3.shape), tuple(go3.shape)))
    xpk = torch.empty(nx, dtype=torch.uint8, device=x3.device)
    gpk = torch.empty(ng, dtype=torch.uint8, device=x3.device)
    check(L.gdm_wgrad_pack_x1_hip(x3.data_ptr(), B, Cin, P, xpk.data_ptr(), _stream()), "gdm_wgrad_pack_x1_hip")
    check(L.gdm_wgrad_pack_go_hip(go3.data_ptr(), B, Cout, P // 32, 32, gpk.data_ptr(), _stream()), "gdm_wgrad_pack_go_hip")
    nchunk = B * P // 128
    parts = _wgrad_parts(nchunk, (Cin // 256) * ((Cout + 127) // 128))
    n = nchunk // parts
    coutp = (Cout + 127) // 128 * 128
    out = torch.empty((parts, Cout, Cin), dtype=torch.float32, device=x3.device)
    check(L.gdm_conv1x1_packed_wb_hip(xpk.data_ptr(), gpk.data_ptr(), n * coutp * 512, parts, 128 * n, Cout, 1, Cin, out.data_ptr(), _stream()),
          "gdm_conv1x1_packed_wb_hip")
    return out[0] if parts == 1 else out.sum(0)


def wgrad_direct_supported(x3, go3, bias=False):
    """Small-channel, many-pixel products: the HBM-bound form that reads the fp32 rows once (include/gdm.h gdm_wgrad_direct_hip).
    Kernel times at batch 24 (tools/bench_wgrad_direct.py under rocprofv3, own vs the fp32 batched GEMM; + 6-12 us for the sum of the
    partials on both sides): 32x32 @ 65536 px 90 vs 320 us, 16x32 48 vs 166, 64x32 135 vs 167, 64x64 @ 16384 px 52 vs 57,
    32x64 @ 4096 12 vs 23, 128x64 @ 16384 112 vs 103, 128x128 @ 4096 48 vs 41 -- so up to 64 x 64 channels, and up to 128 x 128 when the
    bias gradient rides along (its separate reduction costs 34-490 us)."""
    B, Cin, P = x3.shape
    Cout = go3.shape[1]
    lim = 128 if bias else 64
    return (settings.USE_DIRECT_WGRAD and x3.is_cuda and x3.dtype == torch.float32 and go3.dtype == torch.float32 and P % 32 == 0
            and Cin <= lim and Cout <= lim and B * P >= 16384 and _rows_ok(x3) and _rows_ok(go3))


def _rows_ok(t):
    """[B,C,P] with contiguous rows of P floats, channel stride P, any batch stride (a channel slice of a concatenation qualifies)."""
    return t.stride(2) == 1 and t.stride(1) == t.shape[2] and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


def wgrad_direct(x3, go3, bias=False):
    """(dW f32[Cout,Cin], db f32[Cout] or None) = (sum_{b,p} go3[b,:,p] x3[b,:,p]^T, sum_{b,p} go3[b,:,p]) in one pass over the fp32
    rows (split-bf16 MFMA, K split over workgroups, partials added in fixed order)."""
    B, Cin, P = x3.shape
    Cout = go3.shape[1]
    nsteps = B * (P // 32)
    bm, bn = (64 if Cout > 32 else 32), (64 if Cin > 32 else 32)        # the workgroup's block of dW (csrc/gdm_wgrad.hip)
    blocks = -(-Cout // bm) * -(-Cin // bn)
    slices = 4 // ((2 if Cout > 32 else 1) * (2 if Cin > 32 else 1))    # pixel slices inside a workgroup
    nsplit = max(1, min(nsteps // (8 * slices), 512 // blocks))           # >= 8 steps per wave, two workgroups per CU = one resident round
    part = torch.empty((nsplit, Cout, Cin), dtype=torch.float32, device=x3.device)
    bpart = torch.empty((nsplit, Cout), dtype=torch.float32, device=x3.device) if bias else None
    check(_lib.lib().gdm_wgrad_direct_hip(go3.data_ptr(), go3.stride(0), x3.data_ptr(), x3.stride(0), B, Cout, Cin, P, nsplit,
                                          part.data_ptr(), bpart.data_ptr() if bias else None, _stream()), "gdm_wgrad_direct_hip")
    return part.sum(0), (bpart.sum(0) if bias else None)


def _gemm_fwd_mfma_ok(cin, cout, n, B):
    return settings.USE_MFMA_GEMM_TRAIN and gemm_supported(cin, cout, n) and 2.0 * B * n * cin * cout >= 2e9


class _WxTrain(torch.autograd.Function):
    """y[b] = W . x[b] under autograd with the three products on the split-bf16 MFMA GEMM where they are large (>= 2 GFLOP: the tap
    GEMMs of PSPUpsample, the PSP bottleneck, the 512 / 1024-channel fusion convolutions), on hipBLASLt fp32 batched GEMMs otherwise:
    forward W . x, input gradient W^T . go, weight gradient sum_b go[b] . x[b]^T (pixels as the contraction axis)."""

    @staticmethod
    def forward(ctx, x3, w2):
        ctx.save_for_backward(x3, w2)
        B, Cin, n = x3.shape
        Cout = w2.shape[0]
        if x3.is_cuda and _gemm_fwd_mfma_ok(Cin, Cout, n, B):
            return gemm_bf16x3(x3, gemm_pack_weight(w2), Cout)
        if settings.USE_POINTWISE_TRAIN and x3.is_cuda:
            return pointwise([x3], w2, w_rowmajor=True)
        return torch.bmm(w2.unsqueeze(0).expand(B, -1, -1), x3)

    @staticmethod
    def backward(ctx, go):
        x3, w2 = ctx.saved_tensors
        B, Cin, n = x3.shape
        Cout = w2.shape[0]
        go = go.contiguous()
        gx = gw = None
        if ctx.needs_input_grad[0]:
            if x3.is_cuda and _gemm_fwd_mfma_ok(Cout, Cin, n, B):
                gx = gemm_bf16x3(go, conv_pack_weight_dgrad(w2), Cin)
            elif settings.USE_POINTWISE_TRAIN and x3.is_cuda:
                gx = pointwise([go], w2)
            else:
                gx = torch.bmm(w2.t().unsqueeze(0).expand(B, -1, -1), go)
        if ctx.needs_input_grad[1]:
            if gemm_wgrad_supported(x3, go):
                gw = gemm_wgrad(x3, go)
            elif wgrad_direct_supported(x3, go):
                gw = wgrad_direct(x3, go)[0]
            else:
                gw = torch.bmm(go, x3.transpose(1, 2)).sum(0)
        return gx, gw


class _Conv1x1Train(torch.autograd.Function):
    """y[b] = W . x[b] (+ bias) for a 1x1 convolution, with all three products as batched GEMMs on the NCHW tensors as they lie:
    forward W . x, input gradient W^T . go, weight gradient sum_b go[b] . x[b]^T.  torch's convolution backward sends a 1x1
    convolution to MIOpen's implicit-GEMM wgrad / bwd-data kernels, which want NHWC and pay a transposing copy of every operand
    (181 `batched_transpose` launches, 2.7 ms of a 56 ms training step) around fp32 kernels slower than the GEMM library's."""

    @staticmethod
    def forward(ctx, x, w2, bias):
        B, Cin = x.shape[0], x.shape[1]
        x3 = x.reshape(B, Cin, -1)
        ctx.save_for_backward(x3, w2)
        ctx.xshape = x.shape
        ctx.has_bias = bias is not None
        y = torch.empty((B, w2.shape[0]) + tuple(x.shape[2:]), dtype=x.dtype, device=x.device)   # returned as it is (not a view: the
        y3 = y.view(B, w2.shape[0], -1)                                                            # modules' in-place activations follow)
        if settings.USE_POINTWISE_TRAIN and x3.is_cuda and x3.is_contiguous():
            # own exact-fp32 MFMA kernel, the module's [Cout, Cin] weight read in place, the bias in its epilogue
            pointwise([x3], w2, None, bias, ACT_NONE, 0.0, out=y3, w_rowmajor=True)
            return y
        torch.bmm(w2.unsqueeze(0).expand(B, -1, -1), x3, out=y3)
        if bias is not None:
            y3 += bias.view(1, -1, 1)
        return y

    @staticmethod
    def backward(ctx, go):
        x3, w2 = ctx.saved_tensors
        B = x3.shape[0]
        go3 = go.reshape(B, w2.shape[0], -1)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            if settings.USE_POINTWISE_TRAIN and go3.is_cuda:
                gx = pointwise([go3.contiguous()], w2).view(ctx.xshape)     # W^T . go: the same weight read as [K = Cout][Cin]
            else:
                gx = torch.bmm(w2.t().unsqueeze(0).expand(B, -1, -1), go3).view(ctx.xshape)
        want_gb = ctx.has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1] and wgrad_direct_supported(x3, go3, bias=want_gb):
            gw, gb = wgrad_direct(x3, go3, bias=want_gb)             # weight and bias gradient from one read of go
            return gx, gw, gb
        if ctx.needs_input_grad[1]:
            gw = torch.bmm(go3, x3.transpose(1, 2)).sum(0)            # [B, Cout, Cin] partials: <= 25 MB for every layer but one
        if want_gb:
            gb = go3.sum((0, 2))
        return gx, gw, gb


def conv1x1_train_supported(conv, x):
    return (settings.USE_GEMM_CONV1X1_TRAIN and x.is_cuda and x.dtype == torch.float32 and all(k == 1 for k in conv.kernel_size)
            and all(st == 1 for st in conv.stride) and all(p == 0 for p in conv.padding) and conv.groups == 1 and x.dim() in (3, 4))


def conv1x1_train(conv, x):
    """Differentiable 1x1 convolution of an nn.Conv1d / nn.Conv2d module as batched GEMMs (see _Conv1x1Train); large layers take the
    split-bf16 MFMA GEMM for all three products (_WxTrain)."""
    w = conv.weight
    B, Cin = x.shape[0], x.shape[1]
    n = x.numel() // (B * Cin)
    Cout = w.shape[0]
    if x.is_cuda and (_gemm_fwd_mfma_ok(Cin, Cout, n, B) or _gemm_fwd_mfma_ok(Cout, Cin, n, B)):
        y = _WxTrain.apply(x.reshape(B, Cin, n).contiguous(), w.reshape(Cout, Cin).contiguous())
        if conv.bias is not None:
            y = y + conv.bias.view(1, -1, 1)
        return y.view(B, Cout, *x.shape[2:])
    return _Conv1x1Train.apply(x.contiguous(), w.reshape(w.shape[0], w.shape[1]), conv.bias)


def upconv_train_supported(B, cout):
    return B * 9 * cout <= 65535


def psp_combine(g, ys, bias, packed=False):
    """out = relu(g + bias + sum_k bilinear_up(ys[k])); g f32[B,C,H,W], ys four f32[B,C,s,s].  Inference only, in place on g.
    packed: the kernel also writes the packed split-bf16 operand of the GEMMs that read the map next (hung on the result as
    `_gdm_packed`, see gemm_bf16x3_map) -- no pack launch in front of them."""
    g = _dev(g, torch.float32, "g")
    B, C, H, W = g.shape
    ys = [_dev(y, torch.float32, "y") for y in ys]
    assert len(ys) == 4
    opk = None
    if packed and (C == 64 or C % 128 == 0) and W % 32 == 0 and (B * H * W) % 256 == 0 and B * C // 8 <= 65535:
        opk = PackedAct(_packed_buffer(B, C, H, W, g.device), (B, C, H, W))
    check(_lib.lib().gdm_psp_combine2_hip(g.data_ptr(), ys[0].data_ptr(), ys[0].shape[2], ys[1].data_ptr(), ys[1].shape[2],
                                          ys[2].data_ptr(), ys[2].shape[2], ys[3].data_ptr(), ys[3].shape[2],
                                          bias.data_ptr() if bias is not None else None, B, C, H, W, g.data_ptr(),
                                          opk.buf.data_ptr() if opk is not None else None, _stream()), "gdm_psp_combine2_hip")
    if opk is not None:
        g._gdm_packed = opk
    return g


def gather_add_affine_act(x, t, idx, scale, shift, act=ACT_NONE, slope=0.0, hw=None):
    """y[b,c,j] = act(scale[c]*(x[b,c,j] + t[b,c,idx[b,j]]) + shift[c]); x f32[B,C,m], t f32[B,C,n], idx int[B,m(,1)].
    Inference only, in place on x.  With hw = (H, W) of the pixel map (m = H*W) the kernel also writes the packed split-bf16 operand
    of the next convolution / GEMM over it; the caller hangs the returned PackedAct on the map it hands on (`_gdm_packed`)."""
    x = _dev(x, torch.float32, "x")
    t = _dev(t, torch.float32, "t")
    idx = _idx32(idx, "idx")
    B, C, m = x.shape
    n = t.shape[2]
    opk = None
    if hw is not None and hw[0] * hw[1] == m and (C == 64 or C % 128 == 0) and hw[1] % 32 == 0 and (B * m) % 256 == 0:
        opk = PackedAct(_packed_buffer(B, C, hw[0], hw[1], x.device), (B, C, hw[0], hw[1]))
    check(_lib.lib().gdm_gather_add_affine_act2_hip(x.data_ptr(), t.data_ptr(), idx.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                    B, C, n, m, act, float(slope), x.data_ptr(), opk.buf.data_ptr() if opk is not None else None,
                                                    hw[1] if opk is not None else 0, _stream()),
          "gdm_gather_add_affine_act2_hip")
    return (x, opk) if hw is not None else x


def lfa_stage(xyz, idx, feat, w1t, s1, b1, w2t, s2, b2, wft, wmt, sm, bm, slope=0.2):
    """One attentive-pooling stage of RandLA's local feature aggregation in one launch (see include/gdm.h gdm_lfa_stage_hip).
    xyz f32[B,n,3], idx int[B,n,16], feat f32[B,D/2,n(,1)] -> f32[B,OUT,n].  Inference only."""
    xyz = _dev(xyz, torch.float32, "xyz")
    feat = _dev(feat, torch.float32, "feat")
    idx = _idx32(idx, "idx")
    B, n, K = idx.shape
    H = feat.shape[1]
    D, OUT = 2 * H, wmt.shape[1]
    out = torch.empty((B, OUT, n), dtype=torch.float32, device=feat.device)
    z = 0
    check(_lib.lib().gdm_lfa_stage_hip(xyz.data_ptr(), idx.data_ptr(), feat.data_ptr(), w1t.data_ptr(), s1.data_ptr(), b1.data_ptr(),
                                       w2t.data_ptr() if w2t is not None else z, s2.data_ptr() if w2t is not None else z,
                                       b2.data_ptr() if w2t is not None else z, wft.data_ptr(), wmt.data_ptr(), sm.data_ptr(),
                                       bm.data_ptr(), B, n, K, D, OUT, float(slope), out.data_ptr(), _stream()), "gdm_lfa_stage_hip")
    return out


def lfa_supported(d_out, K):
    return K == 16 and d_out in (32, 64, 128, 256)


def conv64_gather_add_act_mfma(x, wpk, t, idx, scale, shift, act=ACT_NONE, slope=0.0, pixel_major=False, t_point_major=False, hw=None):
    """conv1x1_gather_add_act with the 64 x 64 channel mix on split-bf16 MFMA: wpk = pack_rows64(W[64,64]) (row = output channel).
    t f32[B,64,n], or f32[B,n,64] with t_point_major (the gathered term is then one contiguous row per pixel).  Inference only.
    hw = (H, W) of the pixel map (NCHW form): the kernel also writes the packed operand of the next convolution (`_gdm_packed`)."""
    x = _dev(x, torch.float32, "x")
    t = _dev(t, torch.float32, "t")
    idx = _idx32(idx, "idx")
    B, C, m = x.shape
    n = t.shape[1] if t_point_major else t.shape[2]
    if C != 64 or t.shape[2 if t_point_major else 1] != 64 or wpk.numel() != 64 * 256:
        raise ValueError("conv64_gather_add_act_mfma: built for 64 -> 64 channels, got x %s t %s" % (tuple(x.shape), tuple(t.shape)))
    y = torch.empty((B, m, C), dtype=torch.float32, device=x.device) if pixel_major else torch.empty_like(x)
    opk = None
    if hw is not None and not pixel_major and hw[0] * hw[1] == m and hw[1] % 32 == 0 and (B * m) % 256 == 0:
        opk = PackedAct(_packed_buffer(B, C, hw[0], hw[1], x.device), (B, C, hw[0], hw[1]))
    check(_lib.lib().gdm_conv64_gather_add_act_mfma2_hip(x.data_ptr(), wpk.data_ptr(), t.data_ptr(), idx.data_ptr(), scale.data_ptr(),
                                                         shift.data_ptr(), B, n, m, act, float(slope), int(bool(pixel_major)),
                                                         int(bool(t_point_major)), y.data_ptr(), opk.buf.data_ptr() if opk is not None else None,
                                                         int(hw[1]) if opk is not None else 0, _stream()), "gdm_conv64_gather_add_act_mfma2_hip")
    if opk is not None:
        y._gdm_packed = opk
    return y


def conv1x1_gather_add_act(x, wt, t, idx, scale, shift, act=ACT_NONE, slope=0.0, pixel_major=False):
    """y[b,co,j] = act(scale[co]*(sum_ci W[co,ci] x[b,ci,j] + t[b,co,idx[b,j]]) + shift[co]) in one pass for the 64-channel fusion
    levels.  x f32[B,64,m], wt f32[64,64] = W transposed (contiguous), t f32[B,64,n], idx int[B,m(,1)].  Inference only.
    pixel_major: y comes back as f32[B,m,64] (a 256-byte row per pixel: what upconv_final_points reads)."""
    x = _dev(x, torch.float32, "x")
    t = _dev(t, torch.float32, "t")
    wt = _dev(wt, torch.float32, "wt")
    idx = _idx32(idx, "idx")
    B, C, m = x.shape
    if C != 64 or tuple(wt.shape) != (64, 64) or t.shape[1] != 64:
        raise ValueError("conv1x1_gather_add_act: built for 64 -> 64 channels, got x %s wt %s t %s" % (tuple(x.shape), tuple(wt.shape), tuple(t.shape)))
    y = torch.empty((B, m, C), dtype=torch.float32, device=x.device) if pixel_major else torch.empty_like(x)
    check(_lib.lib().gdm_conv1x1_gather_add_act2_hip(x.data_ptr(), wt.data_ptr(), t.data_ptr(), idx.data_ptr(), scale.data_ptr(),
                                                     shift.data_ptr(), B, C, t.shape[2], m, act, float(slope), int(bool(pixel_major)),
                                                     y.data_ptr(), _stream()), "gdm_conv1x1_gather_add_act_hip")
    return y


def point_heads(a, b, layers, last, feat_layer, res_layer, residual=None):
    """The per-point 1x1-convolution chain of GeoMatch.forward in one launch (inference).  a f32[B,Ca,N] (+ b f32[B,128-Ca,N] or None);
    layers = [(wpk, scale|None, shift|None, act)] of 128 -> 128 layers (wpk from gemm_pack_weight, act ACT_NONE / ACT_RELU);
    last = (wpk, bias|None, c_last) or None (the chain ends with its hidden layers).  residual = (ra, rb|None): the tensor added at
    res_layer (default: the input).  -> (out_feat f32[B,128,N] = output of layer feat_layer or None, out_last f32[B,c_last,N] or None)."""
    import ctypes
    a = _dev(a, torch.float32, "a")
    B, Ca, N = a.shape
    if b is not None:
        b = _dev(b, torch.float32, "b")
        if b.shape[0] != B or b.shape[2] != N or b.shape[1] + Ca != 128:
            raise ValueError("point_heads: a %s + b %s must make 128 channels" % (tuple(a.shape), tuple(b.shape)))
    elif Ca != 128:
        raise ValueError("point_heads: a must have 128 channels when b is None, got %d" % Ca)
    n = len(layers)
    vp, fp = ctypes.c_void_p, ctypes.c_void_p
    w_arr = (vp * n)(*[l[0].data_ptr() for l in layers])
    sc_arr = (fp * n)(*[(l[1].data_ptr() if l[1] is not None else None) for l in layers])
    sh_arr = (fp * n)(*[(l[2].data_ptr() if l[2] is not None else None) for l in layers])
    act_arr = (ctypes.c_int * n)(*[int(l[3]) for l in layers])
    for l in layers:
        if l[3] not in (ACT_NONE, ACT_RELU):
            raise ValueError("point_heads: activations are none / ReLU")
    wl, bl, c_last = last if last is not None else (None, None, 0)
    out_feat = torch.empty((B, 128, N), dtype=torch.float32, device=a.device) if feat_layer >= 0 else None
    out_last = torch.empty((B, int(c_last), N), dtype=torch.float32, device=a.device) if c_last else None
    ra, rb = (None, None)
    if residual is not None:
        ra = _dev(residual[0], torch.float32, "residual")
        rb = _dev(residual[1], torch.float32, "residual") if residual[1] is not None else None
        if ra.shape[0] != B or ra.shape[2] != N or ra.shape[1] + (rb.shape[1] if rb is not None else 0) != 128:
            raise ValueError("point_heads: the residual source must make 128 channels of the same points")
    check(_lib.lib().gdm_point_heads2_hip(a.data_ptr(), b.data_ptr() if b is not None else None, Ca,
                                          ra.data_ptr() if ra is not None else None, rb.data_ptr() if rb is not None else None,
                                          ra.shape[1] if ra is not None else 0, B, N, n, w_arr, sc_arr, sh_arr, act_arr,
                                          int(feat_layer), int(res_layer), wl.data_ptr() if wl is not None else None,
                                          bl.data_ptr() if bl is not None else None, int(c_last),
                                          out_feat.data_ptr() if out_feat is not None else None,
                                          out_last.data_ptr() if out_last is not None else None, _stream()), "gdm_point_heads2_hip")
    return out_feat, out_last


def pack_rows64(w2d):
    """f32[R,64] -> R packed split-bf16 rows of 256 B (64 bf16 hi | 64 bf16 lo), u8 tensor."""
    w = _dev(w2d.detach(), torch.float32, "w")
    if w.dim() != 2 or w.shape[1] != 64:
        raise ValueError("pack_rows64: expected [R,64], got %s" % (tuple(w.shape),))
    out = torch.empty(w.shape[0] * 256, dtype=torch.uint8, device=w.device)
    check(_lib.lib().gdm_pack_rows64_hip(w.data_ptr(), w.shape[0], out.data_ptr(), _stream()), "gdm_pack_rows64_hip")
    return out


def upconv_final_points(x_pm, hw, choose, wpk, scale, shift, act, slope, wf_pk, fbias, out_size):
    """The last image stage at the sampled pixels (inference): log_softmax(Wf . act(scale * conv3x3(up(x)) + shift) + fbias) at pixel
    choose[b,n] of the out_size map -> f32[B,64,N].  x_pm f32[B,H*W,64] pixel-major, hw = (H, W); wpk from
    upconv_fused64_pack_weight, wf_pk from pack_rows64(final_weight[64,64])."""
    x_pm = _dev(x_pm, torch.float32, "x_pm")
    choose = _idx32(choose, "choose")
    B = x_pm.shape[0]
    H, W = int(hw[0]), int(hw[1])
    if tuple(x_pm.shape) != (B, H * W, 64):
        raise ValueError("upconv_final_points: x_pm must be [B,H*W,64] with (H, W) = %s, got %s" % ((H, W), tuple(x_pm.shape)))
    choose = choose.reshape(B, -1)
    N = choose.shape[1]
    OH, OW = int(out_size[0]), int(out_size[1])
    out = torch.empty((B, 64, N), dtype=torch.float32, device=x_pm.device)
    check(_lib.lib().gdm_upconv_final_points_hip(x_pm.data_ptr(), choose.data_ptr(), wpk.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                                                 act, float(slope), wf_pk.data_ptr(), fbias.data_ptr() if fbias is not None else None,
                                                 B, H, W, OH, OW, N, out.data_ptr(), _stream()), "gdm_upconv_final_points_hip")
    return out




def stem_pack_weight(weight):
    """w f32[64,3,7,7] -> fragment-ordered split-bf16 weights of the stem kernel (u8 tensor); cache it per weight version."""
    w = _dev(weight.detach(), torch.float32, "weight")
    if tuple(w.shape) != (64, 3, 7, 7):
        raise ValueError("stem_pack_weight: the stem kernel is built for a [64,3,7,7] filter, got %s" % (tuple(w.shape),))
    L = _lib.lib()
    wpk = torch.empty(L.gdm_stem_weight_bytes(), dtype=torch.uint8, device=w.device)
    check(L.gdm_stem_pack_weight_hip(w.data_ptr(), wpk.data_ptr(), _stream()), "gdm_stem_pack_weight_hip")
    return wpk


def stem(x, wpk, scale, shift, packed=True):
    """maxpool3x3/2(relu(scale * conv7x7/2(x) + shift)) in one launch (include/gdm.h gdm_stem_hip).  x f32[B,3,H,W] ->
    f32[B,64,PH,PW]; with `packed` the result also carries (`_gdm_packed`) the split-bf16 operand of the next 3x3 convolution."""
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    if C != 3:
        raise ValueError("stem: 3 input channels, got %d" % C)
    PH, PW = ((H - 1) // 2) // 2 + 1, ((W - 1) // 2) // 2 + 1
    out = torch.empty((B, 64, PH, PW), dtype=torch.float32, device=x.device)
    opk = None
    if packed and (B * PH * PW) % 256 == 0 and PW % 16 == 0:
        opk = PackedAct(_packed_buffer(B, 64, PH, PW, x.device), (B, 64, PH, PW))
    check(_lib.lib().gdm_stem_hip(x.data_ptr(), wpk.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, H, W, out.data_ptr(),
                                  opk.buf.data_ptr() if opk is not None else None, _stream()), "gdm_stem_hip")
    if opk is not None:
        out._gdm_packed = opk
    return out


def affine_relu_maxpool(x, scale, shift):
    """MaxPool2d(3, 2, 1)(relu(scale[c] * x + shift[c])) in one pass (the ResNet stem behind conv1).  Inference only."""
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    y = torch.empty((B, C, (H - 1) // 2 + 1, (W - 1) // 2 + 1), dtype=torch.float32, device=x.device)
    check(_lib.lib().gdm_affine_relu_maxpool_hip(x.data_ptr(), scale.data_ptr(), shift.data_ptr(), B, C, H, W, y.data_ptr(), _stream()),
          "gdm_affine_relu_maxpool_hip")
    return y


def conv1x1_logsoftmax(x, weight, bias):
    """log_softmax over channels of a 64->64 1x1 convolution, one pass (the `final` stage, pspnet.py:108-112). Inference only."""
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    # W^T (the kernel reads 64 contiguous scalars per ci), cached ON the weight tensor: it lives and dies with it.  (A module-level dict
    # keyed by id(weight) served a NEW tensor that got a dead one's id, address and version the old one's transpose: wrong output, found
    # by a test that builds several weights in a row.)
    key = (weight._version, weight.data_ptr())
    cache = getattr(weight, "_gdm_final_wt", None)
    if cache is None or cache[0] != key:
        cache = (key, weight.detach().reshape(C, C).t().contiguous())
        weight._gdm_final_wt = cache
    w = cache[1]
    out = torch.empty_like(x)
    check(_lib.lib().gdm_conv1x1_logsoftmax_hip(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                                B, C, H * W, out.data_ptr(), _stream()), "gdm_conv1x1_logsoftmax_hip")
    return out


def channel_sum(t):
    """t f32[B,C,...] -> f32[C] sums over batch and inner dimensions (bias gradients): the streaming reduction of the BatchNorm kernels
    (fp32 lanes, double above) where its shape rules hold -- torch's generic reduce_kernel moves these maps at 0.7 TB/s."""
    B, C = t.shape[0], t.shape[1]
    inner = t.numel() // max(B * C, 1)
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and B * C <= 65535 and inner >= 4 and inner % 4 == 0
            and t.data_ptr() % 16 == 0):
        return t.sum(dim=[0] + list(range(2, t.dim())))
    L = _lib.lib()
    sums = torch.empty(L.gdm_bn_sums_len(B, C, inner), dtype=torch.float64, device=t.device)
    check(L.gdm_bn_stats_hip(t.data_ptr(), B, C, inner, sums.data_ptr(), _stream()), "gdm_bn_stats_hip")
    return sums[:-2].view(-1, C, 2)[:, :, 0].sum(0).float()


class _PspPools(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = _dev(x, torch.float32, "x")
        B, C, H, W = x.shape
        outs = [torch.empty((B, C, s_, s_), dtype=torch.float32, device=x.device) for s_ in (1, 2, 3, 6)]
        check(_lib.lib().gdm_psp_pools_hip(x.data_ptr(), B * C, H, W, outs[0].data_ptr(), outs[1].data_ptr(), outs[2].data_ptr(),
                                           outs[3].data_ptr(), _stream()), "gdm_psp_pools_hip")
        ctx.shape = (B, C, H, W)
        return tuple(outs)

    @staticmethod
    def backward(ctx, g1, g2, g3, g6):
        B, C, H, W = ctx.shape
        gs = [_dev(g, torch.float32, "grad") for g in (g1, g2, g3, g6)]
        gx = torch.empty((B, C, H, W), dtype=torch.float32, device=gs[0].device)
        check(_lib.lib().gdm_psp_pools_bwd_hip(gs[0].data_ptr(), gs[1].data_ptr(), gs[2].data_ptr(), gs[3].data_ptr(), B * C, H, W,
                                               gx.data_ptr(), _stream()), "gdm_psp_pools_bwd_hip")
        return gx


def psp_pools_supported(h, w):
    return 36 <= h * w <= 4096 and h >= 6 and w >= 6


def psp_pools(x):
    """Adaptive average pools to 1,2,3,6 bins in one pass: x f32[B,C,H,W] -> four f32[B,C,s,s] (differentiable)."""
    return list(_PspPools.apply(x))


class _PspCombine(torch.autograd.Function):
    """out = relu(g + bias + sum_k bilinear_up(ys[k])) with its backward: the training form of psp_combine."""

    @staticmethod
    def forward(ctx, g, bias, y1, y2, y3, y4):
        g = _dev(g, torch.float32, "g")
        ys = [_dev(y, torch.float32, "y") for y in (y1, y2, y3, y4)]
        B, C, H, W = g.shape
        out = torch.empty_like(g)
        check(_lib.lib().gdm_psp_combine_hip(g.data_ptr(), ys[0].data_ptr(), ys[0].shape[2], ys[1].data_ptr(), ys[1].shape[2],
                                             ys[2].data_ptr(), ys[2].shape[2], ys[3].data_ptr(), ys[3].shape[2],
                                             bias.data_ptr() if bias is not None else None, B, C, H, W, out.data_ptr(), _stream()),
              "gdm_psp_combine_hip")
        ctx.save_for_backward(out)
        ctx.sizes = [y.shape[2] for y in ys]
        ctx.has_bias = bias is not None
        return out

    @staticmethod
    def backward(ctx, go):
        (out,) = ctx.saved_tensors
        B, C, H, W = out.shape
        gpre = torch.where(out > 0, go, torch.zeros((), dtype=go.dtype, device=go.device)).contiguous()
        gys = []
        L = _lib.lib()
        for s_ in ctx.sizes:
            gy = torch.empty((B, C, s_, s_), dtype=torch.float32, device=go.device)
            check(L.gdm_upsample_bilinear_bwd_hip(gpre.data_ptr(), B * C, s_, s_, H, W, gy.data_ptr(), _stream()), "gdm_upsample_bilinear_bwd_hip")
            gys.append(gy)
        gb = channel_sum(gpre) if ctx.has_bias else None
        return gpre, gb, gys[0], gys[1], gys[2], gys[3]


def psp_combine_train(g, ys, bias):
    """Differentiable psp_combine (not in place): relu(g + bias + sum_k bilinear_up(ys[k]))."""
    assert len(ys) == 4
    return _PspCombine.apply(g, bias, ys[0], ys[1], ys[2], ys[3])


class BufferPool:
    """Owner-held scratch buffers of the split-bf16 kernels (zero-bordered packed activation maps, matching workspaces).  Whoever
    captures a step in a hipGraph owns one (infer.GraphedPipeline, train_graph.GraphedTrainStep, bench.py's step) and enters it
    with `ops.buffer_pool(pool)` around warm-up, capture and any later eager call of the same step: the buffers -- and the zero
    fill of their borders, captured once per buffer -- then belong to that graph's private memory pool and to nothing else.
    Eager code outside any scope uses the process-wide default pool."""

    def __init__(self):
        self.packed = {}
        self.workspace = {}


_default_pool = BufferPool()
_pool = _default_pool
_warned_capture = []


class buffer_pool:
    def __init__(self, pool):
        self.pool, self.prev = pool, None

    def __enter__(self):
        global _pool
        self.prev, _pool = _pool, self.pool
        return self.pool

    def __exit__(self, *exc):
        global _pool
        _pool = self.prev
        return False


def _capturing_unscoped():
    """A hipGraph capture that runs outside any buffer_pool scope: its scratch buffers must not enter (or come from) the
    process-wide pool -- they would live in this graph's private memory pool, and a later capture of the same shape would
    silently write into them -- so they are allocated per call (each zero fill is then replayed with the graph: slower, correct)."""
    if _pool is _default_pool and torch.cuda.is_current_stream_capturing():
        if not _warned_capture:
            _warned_capture.append(1)
            import warnings
            warnings.warn("geometric_aware_dense_matching_amd.ops: hipGraph capture outside ops.buffer_pool(...): scratch buffers "
                          "are allocated per call inside the graph; give the capture a BufferPool of its own")
        return True
    return False




def conv3x3_wgrad_supported(x, go):
    """The split-bf16 MFMA weight-gradient path: 32- or 64-wide maps, Cin a multiple of 256 (the per-part weights of the GEMM want
    whole 256-row tiles per part) -- the 32 x 32 half of the trunk, where torch's path (MIOpen fp32 implicit GEMM) is 2.5x slower."""
    B, Cin, H, W = x.shape
    return (settings.USE_MFMA_WGRAD and x.is_cuda and x.dtype == torch.float32 and go.shape[0] == B and tuple(go.shape[2:]) == (H, W)
            and W in (32, 64) and (H * W) % 128 == 0 and Cin % 256 == 0 and go.shape[1] % 8 == 0)


def conv3x3_wgrad(x, go, parts=None):
    """dW f32[Cout,Cin,3,3] of a 3x3 / stride 1 / pad 1 convolution from its input x f32[B,Cin,H,W] and output gradient go
    f32[B,Cout,H,W], on the split-bf16 MFMA GEMM (include/gdm.h gdm_wgrad_*): the contraction runs over the B*H*W pixels, split
    into `parts` equal parts (one launch), whose [Cout,9,Cin] partial products are added here in a fixed order (deterministic)."""
    x = _dev(x, torch.float32, "x")
    go = _dev(go, torch.float32, "grad_out")
    B, Cin, H, W = x.shape
    Cout = go.shape[1]
    L = _lib.lib()
    nx, ng = L.gdm_wgrad_x_bytes(B, Cin, H, W), L.gdm_wgrad_go_bytes(B, Cout, H, W)
    if nx == 0 or ng == 0 or (9 * Cin) % 256 != 0:
        raise ValueError("conv3x3_wgrad: unsupported shape x %s grad_out %s" % (tuple(x.shape), tuple(go.shape)))
    xpk = torch.empty(nx, dtype=torch.uint8, device=x.device)
    gpk = torch.empty(ng, dtype=torch.uint8, device=x.device)
    check(L.gdm_wgrad_pack_x_hip(x.data_ptr(), B, Cin, H, W, xpk.data_ptr(), _stream()), "gdm_wgrad_pack_x_hip")
    check(L.gdm_wgrad_pack_go_hip(go.data_ptr(), B, Cout, H, W, gpk.data_ptr(), _stream()), "gdm_wgrad_pack_go_hip")
    nchunk = B * H * W // 128
    if parts is None:
        # parts: the largest divisor of the chunk count that keeps the launch at about one round of workgroups (a part has
        # 9 Cin / 256 x ceil(Cout / 128) of them) and at least four chunks per part
        tiles = (9 * Cin // 256) * ((Cout + 127) // 128)
        parts = 1
        for d in range(1, nchunk + 1):
            if nchunk % d == 0 and d * tiles <= 320 and nchunk // d >= 4:
                parts = d
    if nchunk % parts != 0:
        raise ValueError("conv3x3_wgrad: %d parts do not divide the %d chunks of the contraction" % (parts, nchunk))
    n = nchunk // parts
    coutp = (Cout + 127) // 128 * 128
    out = torch.empty((parts, Cout, 9, Cin), dtype=torch.float32, device=x.device)
    check(L.gdm_conv1x1_packed_wb_hip(xpk.data_ptr(), gpk.data_ptr(), n * coutp * 512, parts, 128 * n, Cout, 9, Cin, out.data_ptr(), _stream()),
          "gdm_conv1x1_packed_wb_hip")
    dw = out[0] if parts == 1 else out.sum(0)
    return dw.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).contiguous()


def conv3x3_supported(x, weight, stride=(1, 1), padding=(1, 1), dilation=(1, 1)):
    """3x3 / pad 1 / no dilation, stride 1 or 2 (output width a multiple of 32), Cin 64 or a multiple of 128."""
    cin = weight.shape[1]
    st = tuple(stride)
    if st not in ((1, 1), (2, 2)):
        return False
    return (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and tuple(weight.shape[2:]) == (3, 3)
            and tuple(padding) == (1, 1) and tuple(dilation) == (1, 1)
            and x.shape[2] % st[0] == 0 and x.shape[3] % st[1] == 0 and (x.shape[3] // st[1]) % 32 == 0
            and (cin == 64 or cin % 128 == 0) and weight.shape[0] % 8 == 0 and x.shape[0] <= 65535)


def conv_pack_weight_dgrad(weight):
    """The packed weights of the input-gradient convolution of a 3x3/s1/p1 (w f32[Cout,Cin,3,3]) or 1x1 (w f32[Cout,Cin]) layer, straight
    from the forward weight: one launch instead of flip + transpose + contiguous + pack (include/gdm.h gdm_conv_pack_weight_dgrad_hip)."""
    weight = _dev(weight.detach(), torch.float32, "weight")
    Cout, Cin = weight.shape[0], weight.shape[1]
    taps = 9 if weight.dim() == 4 else 1
    L = _lib.lib()
    nbytes = L.gdm_conv3x3_weight_bytes(Cin, Cout) if taps == 9 else L.gdm_conv1x1_weight_bytes(Cin, Cout)
    wpk = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    check(L.gdm_conv_pack_weight_dgrad_hip(weight.data_ptr(), Cout, Cin, taps, wpk.data_ptr(), _stream()), "gdm_conv_pack_weight_dgrad_hip")
    return wpk


def conv3x3_pack_weight(weight):
    """w f32[Cout,Cin,3,3] -> packed split-bf16 rows (u8 tensor); cache it per weight version."""
    weight = _dev(weight.detach(), torch.float32, "weight")
    Cout, Cin = weight.shape[0], weight.shape[1]
    L = _lib.lib()
    wpk = torch.empty(L.gdm_conv3x3_weight_bytes(Cout, Cin), dtype=torch.uint8, device=weight.device)
    check(L.gdm_conv3x3_pack_weight_hip(weight.data_ptr(), Cout, Cin, wpk.data_ptr(), _stream()), "gdm_conv3x3_pack_weight_hip")
    return wpk


class PackedAct:
    """An activation map in the convolution kernel's operand layout (bf16 hi / lo planes of 8 channels, one-pixel zero border):
    what conv3x3_bf16x3(..., out_packed=True) hands to the next convolution instead of a pack launch."""
    __slots__ = ("buf", "shape")

    def __init__(self, buf, shape):
        self.buf, self.shape = buf, tuple(shape)


def _packed_buffer(B, C, H, W, device, avoid=None):
    """Zero-bordered operand buffer for [B,C,H,W] from the current BufferPool (two per shape and stream: a layer reads one and
    writes the other).  Only interior pixels are ever written, so the border stays zero."""
    nbytes = _lib.lib().gdm_conv3x3_act_bytes(B, C, H, W)
    if _capturing_unscoped():
        return torch.zeros(nbytes, dtype=torch.uint8, device=device)
    key = (B, C, H, W, device.index, _stream_lane(device))
    pool = _pool.packed.setdefault(key, [])
    for buf in pool:
        if avoid is None or buf.data_ptr() != avoid.data_ptr():
            return buf
    buf = torch.zeros(nbytes, dtype=torch.uint8, device=device)
    pool.append(buf)
    return buf


def conv3x3_pack_act(x):
    """x f32[B,C,H,W] -> PackedAct (one launch)."""
    x = _dev(x, torch.float32, "x")
    B, C, H, W = x.shape
    buf = _packed_buffer(B, C, H, W, x.device)
    check(_lib.lib().gdm_conv3x3_pack_act_hip(x.data_ptr(), B, C, H, W, buf.data_ptr(), _stream()), "gdm_conv3x3_pack_act_hip")
    return PackedAct(buf, (B, C, H, W))


def conv3x3_bf16x3(x, wpk, cout, scale=None, shift=None, act=ACT_NONE, res=None, out_f32=True, out_packed=False, stride=1):
    """3x3/p1 convolution (stride 1 or 2) of x f32[B,Cin,H,W] (or a PackedAct) with packed weights on split-bf16 MFMA (+ per-channel
    scale/shift, optional residual, optional ReLU).  Returns the fp32 map, or -- out_packed -- (fp32 map or None, PackedAct of the
    result): the epilogue writes the next convolution's operand itself.  Inference only."""
    xp = x if isinstance(x, PackedAct) else conv3x3_pack_act(x)
    B, Cin, H, W = xp.shape
    if stride != 1:
        if stride != 2 or H % 2 or W % 2:
            raise ValueError("conv3x3_bf16x3: stride %r on a %dx%d map" % (stride, H, W))
        H, W = H // 2, W // 2
    L = _lib.lib()
    dev = xp.buf.device
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=dev) if out_f32 else None
    opk = None
    if out_packed:
        if (B * H * W) % 256 != 0 or cout % 8 != 0:
            raise ValueError("conv3x3_bf16x3: packed output needs B*H*W %% 256 == 0 and Cout %% 8 == 0")
        opk = PackedAct(_packed_buffer(B, cout, H, W, dev, avoid=xp.buf), (B, cout, H, W))
    if res is not None:
        res = _dev(res, torch.float32, "res")
        assert tuple(res.shape) == (B, cout, H, W)
    check(L.gdm_conv3x3_strided_hip(xp.buf.data_ptr(), wpk.data_ptr(), scale.data_ptr() if scale is not None else None,
                                    shift.data_ptr() if shift is not None else None, res.data_ptr() if res is not None else None,
                                    B, Cin, cout, H, W, int(stride), act, out.data_ptr() if out is not None else None,
                                    opk.buf.data_ptr() if opk is not None else None, _stream()), "gdm_conv3x3_strided_hip")
    return (out, opk) if out_packed else out


def gemm_bf16x3_map(x, wpk, cout):
    """W @ x over the channels of a map x f32[B,Cin,H,W] -> f32[B,cout,H,W] on split-bf16 MFMA.  If x carries the packed operand its
    producer wrote (`_gdm_packed`, the trunk's residual blocks), the GEMM reads that and no pack launch is needed."""
    B, Cin, H, W = x.shape
    xp = getattr(x, "_gdm_packed", None)
    if isinstance(xp, PackedAct) and xp.shape == (B, Cin, H, W) and W % 32 == 0:
        return conv1x1_packed2d(xp, wpk, cout)
    return gemm_bf16x3(x.reshape(B, Cin, H * W), wpk, cout).view(B, cout, H, W)


def conv1x1_packed2d(xp, wpk, cout, scale=None, shift=None, act=ACT_NONE, stride=1):
    """1x1 convolution (stride 1 or 2) of a PackedAct map with gemm_pack_weight'ed weights -> f32[B,cout,H/stride,W/stride]: the
    downsample branch of a residual block on the packed operand its 3x3 convolution already reads.  Inference only."""
    B, Cin, H, W = xp.shape
    if stride not in (1, 2) or H % stride or W % stride:
        raise ValueError("conv1x1_packed2d: stride %r on a %dx%d map" % (stride, H, W))
    H, W = H // stride, W // stride
    out = torch.empty((B, cout, H, W), dtype=torch.float32, device=xp.buf.device)
    check(_lib.lib().gdm_conv1x1_strided_hip(xp.buf.data_ptr(), wpk.data_ptr(), scale.data_ptr() if scale is not None else None,
                                             shift.data_ptr() if shift is not None else None, B, Cin, cout, H, W, int(stride), act,
                                             out.data_ptr(), _stream()), "gdm_conv1x1_strided_hip")
    return out


class _Conv3x3Train(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        return conv3x3_bf16x3(x, conv3x3_pack_weight(weight), weight.shape[0])

    @staticmethod
    def backward(ctx, go):
        x, weight = ctx.saved_tensors
        go = _dev(go, torch.float32, "grad")
        gx = gw = None
        if ctx.needs_input_grad[0]:
            # dgrad of a 3x3/s1/p1 convolution = the same convolution of grad_out with the flipped, transposed filter
            gx = conv3x3_bf16x3(go, conv_pack_weight_dgrad(weight), weight.shape[1])
        if ctx.needs_input_grad[1]:
            if conv3x3_wgrad_supported(x, go):
                gw = conv3x3_wgrad(x, go)
            else:
                gw = torch.ops.aten.convolution_backward(go, x, weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
        return gx, gw


def conv3x3_train(x, weight):
    """Differentiable 3x3/s1/p1 convolution without bias: forward and input gradient on the split-bf16 MFMA kernel (same error bound as
    the inference kernel, ~1e-5 relative), weight gradient on MIOpen.  Shapes as conv3x3_supported."""
    return _Conv3x3Train.apply(x, weight)


def gemm_supported(cin, cout, npix):
    """K in whole 128-channel chunks (or exactly 64); output channels are padded to 128 inside the kernel (zero weight rows, masked stores), which
    pays once the padding wastes at most a third of the MFMA work."""
    return (cin % 128 == 0 or cin == 64) and (cout % 128 == 0 or cout >= 192) and npix % 32 == 0


def gemm_pack_weight(weight2d):
    """W f32[Cout,Cin] -> packed split-bf16 rows for gemm_bf16x3."""
    w = _dev(weight2d.detach(), torch.float32, "weight")
    Cout, Cin = w.shape
    L = _lib.lib()
    wpk = torch.empty(L.gdm_conv1x1_weight_bytes(Cout, Cin), dtype=torch.uint8, device=w.device)
    check(L.gdm_conv1x1_pack_weight_hip(w.data_ptr(), Cout, Cin, wpk.data_ptr(), _stream()), "gdm_conv1x1_pack_weight_hip")
    return wpk


def gemm_bf16x3(x, wpk, cout, scale=None, shift=None, act=ACT_NONE, pixel_major=False):
    """out[b] = act(scale * (W @ x[b]) + shift) on split-bf16 MFMA.  x f32[B,Cin,n] (channel-major, n % 32 == 0) ->
    f32[B,cout,n], or f32[B*n, cout] when pixel_major.  Inference only."""
    x = _dev(x, torch.float32, "x")
    B, Cin, n = x.shape
    L = _lib.lib()
    xpk = _packed_buffer(B, Cin, 1, n, x.device)
    check(L.gdm_conv3x3_pack_act_hip(x.data_ptr(), B, Cin, 1, n, xpk.data_ptr(), _stream()), "gdm_conv3x3_pack_act_hip")
    out = torch.empty((B * n, cout) if pixel_major else (B, cout, n), dtype=torch.float32, device=x.device)
    check(L.gdm_conv1x1_packed_hip(xpk.data_ptr(), wpk.data_ptr(), scale.data_ptr() if scale is not None else None,
                                   shift.data_ptr() if shift is not None else None, B, Cin, cout, 1, n, act,
                                   1 if pixel_major else 0, out.data_ptr(), _stream()), "gdm_conv1x1_packed_hip")
    return out


def spline_packed_buffer(C, M, device, avoid=None):
    """The zero-bordered operand buffer a SplineConv layer's kernels write for the NEXT layer's grouped GEMM (a [1, C, 1, M] map)."""
    return _packed_buffer(1, C, 1, M, device, avoid=avoid)


def gemm_grouped(x_cm, wpk, rowidx, tile_co0, cout_total, xpk=None):
    """Grouped, gathered GEMM on the split-bf16 MFMA kernel (include/gdm.h gdm_gemm_grouped_hip): x_cm f32[1,Cin,M] (channel-major),
    wpk = gemm_pack_weight of a [cout_total, Cin] matrix, rowidx i32[R] (R % 256 == 0), tile_co0 i32[R/256] -> Y f32[R,128].
    xpk: the packed operand of x_cm when its producer wrote it (gdm_spline_*3_hip); otherwise one pack launch makes it here."""
    x_cm = _dev(x_cm, torch.float32, "x")
    _, Cin, M = x_cm.shape
    R = rowidx.shape[0]
    L = _lib.lib()
    if xpk is None:
        xpk = _packed_buffer(1, Cin, 1, M, x_cm.device)
        check(L.gdm_conv3x3_pack_act_hip(x_cm.data_ptr(), 1, Cin, 1, M, xpk.data_ptr(), _stream()), "gdm_conv3x3_pack_act_hip")
    Y = torch.empty((R, 128), dtype=torch.float32, device=x_cm.device)
    check(L.gdm_gemm_grouped_hip(xpk.data_ptr(), wpk.data_ptr(), rowidx.data_ptr(), tile_co0.data_ptr(), R, M, Cin, cout_total,
                                 Y.data_ptr(), _stream()), "gdm_gemm_grouped_hip")
    return Y


class _UpsampleBilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, OH, OW):
        x = _dev(x, torch.float32, "x")
        B, C, H, W = x.shape
        out = torch.empty((B, C, OH, OW), dtype=torch.float32, device=x.device)
        check(_lib.lib().gdm_upsample_bilinear_hip(x.data_ptr(), B * C, H, W, OH, OW, out.data_ptr(), _stream()),
              "gdm_upsample_bilinear_hip")
        ctx.hw = (H, W)
        return out

    @staticmethod
    def backward(ctx, go):
        go = go.contiguous()
        B, C, OH, OW = go.shape
        H, W = ctx.hw
        g = torch.empty((B, C, H, W), dtype=torch.float32, device=go.device)
        check(_lib.lib().gdm_upsample_bilinear_bwd_hip(go.data_ptr(), B * C, H, W, OH, OW, g.data_ptr(), _stream()),
              "gdm_upsample_bilinear_bwd_hip")
        return g, None, None


class _PReLU1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, slope):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(_lib.lib().gdm_prelu1_hip(x.data_ptr(), slope.data_ptr(), x.numel(), y.data_ptr(), _stream()), "gdm_prelu1_hip")
        ctx.save_for_backward(x, slope)
        return y

    @staticmethod
    def backward(ctx, go):
        x, slope = ctx.saved_tensors
        go = go.contiguous()
        gx = torch.empty_like(x)
        gs = torch.zeros_like(slope)
        check(_lib.lib().gdm_prelu1_bwd_hip(x.data_ptr(), go.data_ptr(), slope.data_ptr(), x.numel(), gx.data_ptr(), gs.data_ptr(), _stream()),
              "gdm_prelu1_bwd_hip")
        return gx, gs


def prelu1(x, slope):
    """Single-parameter PReLU with a streaming HIP backward (training path of PSPUpsample). x f32 cuda, numel % 4 == 0."""
    x = _dev(x, torch.float32, "x")
    return _PReLU1.apply(x, slope)


def upsample_bilinear(x, size):
    """x f32[B,C,H,W] -> f32[B,C,OH,OW], bilinear, align_corners=True (pspnet.py:26-29,38)."""
    return _UpsampleBilinear.apply(x, int(size[0]), int(size[1]))


# --------------------------------------------------------------------------------------
# matching
# --------------------------------------------------------------------------------------
def _workspace(nbytes, device):
    if _capturing_unscoped():
        return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)
    key = (device.index, _stream_lane(device))
    ws = _pool.workspace.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        _pool.workspace[key] = ws
    return ws


def match(scene, model, precision=MATCH_BF16X3, return_sim=False):
    """scene f32[B,128,N] (end_points['rgbd']), model f32[128,M] (end_points['mesh'][0]).
    Returns best_idx i32[B,N], best_sim f32[B,N] (and sim f32[B,N,M] when return_sim)."""
    scene = _dev(scene, torch.float32, "scene")
    model = _dev(model, torch.float32, "model")
    if model.dim() == 3:
        assert model.shape[0] == 1
        model = model[0]
    B, D, N = scene.shape
    M = model.shape[1]
    assert model.shape[0] == D
    L = _lib.lib()
    nbytes = L.gdm_match_workspace_bytes(B, N, M)
    ws = _workspace(nbytes, scene.device)
    best_idx = torch.empty((B, N), dtype=torch.int32, device=scene.device)
    best_sim = torch.empty((B, N), dtype=torch.float32, device=scene.device)
    sim = torch.empty((B, N, M), dtype=torch.float32, device=scene.device) if return_sim else None
    check(L.gdm_match_hip(scene.data_ptr(), model.data_ptr(), B, D, N, M, precision, best_idx.data_ptr(),
                          best_sim.data_ptr(), sim.data_ptr() if return_sim else None, ws.data_ptr(), ws.numel(),
                          _stream()), "gdm_match_hip")
    return (best_idx, best_sim, sim) if return_sim else (best_idx, best_sim)


def match_pack(x, precision=MATCH_BF16X3, out=None):
    """x f32[R,128,n] (or [128,n]) channel-major -> u8[R*n, 512] normalised packed rows (stage 1 of match)."""
    x = _dev(x, torch.float32, "x")
    if x.dim() == 2:
        x = x.unsqueeze(0)
    R, D, n = x.shape
    L = _lib.lib()
    if out is None:
        out = torch.empty((L.gdm_match_rows_bytes(R * n),), dtype=torch.uint8, device=x.device)
    check(L.gdm_match_pack_hip(x.data_ptr(), R, D, n, precision, out.data_ptr(), _stream()), "gdm_match_pack_hip")
    return out


def match_pack2(x1, x2, precision=MATCH_BF16X3):
    """match_pack(x1), match_pack(x2) in one launch (include/gdm.h gdm_match_pack2_hip): the scene and the model descriptors of a step."""
    x1 = _dev(x1, torch.float32, "x1")
    x2 = _dev(x2, torch.float32, "x2")
    x1 = x1.unsqueeze(0) if x1.dim() == 2 else x1
    x2 = x2.unsqueeze(0) if x2.dim() == 2 else x2
    (R1, D, n1), (R2, D2, n2) = x1.shape, x2.shape
    if D != D2:
        raise ValueError("match_pack2: %d vs %d channels" % (D, D2))
    L = _lib.lib()
    o1 = torch.empty((L.gdm_match_rows_bytes(R1 * n1),), dtype=torch.uint8, device=x1.device)
    o2 = torch.empty((L.gdm_match_rows_bytes(R2 * n2),), dtype=torch.uint8, device=x1.device)
    check(L.gdm_match_pack2_hip(x1.data_ptr(), R1, n1, o1.data_ptr(), x2.data_ptr(), R2, n2, o2.data_ptr(), D, precision, _stream()),
          "gdm_match_pack2_hip")
    return o1, o2


def match_packed(scene_rows, model_rows, B, N, M, precision=MATCH_BF16X3, return_sim=False, sim_out=None):
    """Stage 2 of match on packed rows: the MFMA similarity + arg-max kernel (+ split merge)."""
    L = _lib.lib()
    dev = scene_rows.device
    part = _workspace(L.gdm_match_partial_bytes(B, N), dev)
    best_idx = torch.empty((B, N), dtype=torch.int32, device=dev)
    best_sim = torch.empty((B, N), dtype=torch.float32, device=dev)
    sim = None
    if return_sim:
        sim = sim_out if sim_out is not None else torch.empty((B, N, M), dtype=torch.float32, device=dev)
    check(L.gdm_match_packed_hip(scene_rows.data_ptr(), model_rows.data_ptr(), B * N, M, precision, best_idx.data_ptr(),
                                 best_sim.data_ptr(), sim.data_ptr() if return_sim else None, part.data_ptr(), part.numel(),
                                 _stream()), "gdm_match_packed_hip")
    return (best_idx, best_sim, sim) if return_sim else (best_idx, best_sim)


def seg_mask(seg):
    """seg f32[B,2,N] -> (mask u8[B,N] = argmax==1, count i32[B])."""
    seg = _dev(seg, torch.float32, "seg")
    B, two, N = seg.shape
    assert two == 2
    mask = torch.empty((B, N), dtype=torch.uint8, device=seg.device)
    count = torch.empty((B,), dtype=torch.int32, device=seg.device)
    check(_lib.lib().gdm_seg_mask_hip(seg.data_ptr(), B, N, mask.data_ptr(), count.data_ptr(), _stream()),
          "gdm_seg_mask_hip")
    return mask, count
