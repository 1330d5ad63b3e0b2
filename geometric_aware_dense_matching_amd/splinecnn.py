"""Object-model (mesh) embedding: kNN graph (k=4) + Cartesian pseudo-coordinates + 3 SplineConv
layers + concat + Linear(393,128).  Mirrors /root/reference/models/SplineCNN.py:101-251.

torch_geometric / torch_spline_conv / torch_cluster are third-party, un-vendored and absent here
(reference README.md:24-25), so the graph transform and SplineConv are re-stated from their
published definitions (see csrc/gdm_spline.hip and DESIGN.md: parity unpinned for this op):
  KNNGraph(k=4)  : for every vertex its 4 nearest other vertices; edges neighbour -> centre
  Cartesian()    : attr = (pos_j - pos_i) / (2 max|.|) + 0.5 over all edges
  SplineConv     : dim 3, kernel 5^3, degree 1, open, aggr mean, root weight, bias

MI355X formulation: the dense part of SplineConv is one GEMM X @ [W_0|...|W_124]
(hipBLASLt through torch.matmul), the sparse part (basis, 8-row gather, mean, root, bias, ReLU) is
one HIP kernel over CSR edges (ops below).  The graph's kNN runs on the HIP kNN kernel.

The embedding is input independent (`forward()` takes no arguments, SplineCNN.py:234); the
reference nevertheless recomputes it on every GeoMatch.forward (geoMatch.py:179).  `forward`
recomputes too (training needs it); `GeoMatch(cache_mesh_in_eval=True)` may reuse it in eval.
"""
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops, settings
from ._lib import check
from .layers import cached_gemm_weight
from .synthetic import COLOR_MEAN, COLOR_STD_MESH

KERNEL_SIZE = 5


class _SplineAggregate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xw, root, bias, rowptr, src, attr, relu):
        M, nk, C = xw.shape
        out = torch.empty((M, C), dtype=torch.float32, device=xw.device)
        check(_lib.lib().gdm_spline_aggregate_hip(xw.data_ptr(), rowptr.data_ptr(), src.data_ptr(), attr.data_ptr(),
                                                  root.data_ptr(), bias.data_ptr(), M, C, KERNEL_SIZE, int(relu),
                                                  out.data_ptr(), ops._stream()), "gdm_spline_aggregate_hip")
        ctx.save_for_backward(rowptr, src, attr, out)
        ctx.relu = relu
        ctx.nk = nk
        return out

    @staticmethod
    def backward(ctx, go):
        rowptr, src, attr, out = ctx.saved_tensors
        go = go.contiguous()
        if ctx.relu:
            go = go * (out > 0).to(go.dtype)
        M, C = go.shape
        gxw = torch.zeros((M, ctx.nk, C), dtype=torch.float32, device=go.device)
        check(_lib.lib().gdm_spline_aggregate_bwd_hip(go.data_ptr(), rowptr.data_ptr(), src.data_ptr(), attr.data_ptr(),
                                                      M, C, KERNEL_SIZE, gxw.data_ptr(), ops._stream()),
              "gdm_spline_aggregate_bwd_hip")
        return gxw, go, go.sum(dim=0), None, None, None, None


class SplineConv(nn.Module):
    """torch_geometric.nn.SplineConv(in, out, dim=3, kernel_size=5) parameter layout:
    weight [125, in, out], lin.weight [out, in] (root), bias [out]."""

    def __init__(self, cin, cout):
        super().__init__()
        self.cin, self.cout = cin, cout
        self.weight = nn.Parameter(torch.empty(KERNEL_SIZE ** 3, cin, cout))
        self.lin = nn.Linear(cin, cout, bias=False)
        self.bias = nn.Parameter(torch.zeros(cout))
        bound = 1.0 / np.sqrt(cin * KERNEL_SIZE ** 3)
        nn.init.uniform_(self.weight, -bound, bound)
        nn.init.uniform_(self.lin.weight, -bound, bound)

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        old = prefix + "root"                                  # torch_geometric 1.x name: root [in, out]
        if old in state_dict and prefix + "lin.weight" not in state_dict:
            state_dict[prefix + "lin.weight"] = state_dict.pop(old).t().contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _root_t(self):
        w = self.lin.weight
        key = (w._version, w.data_ptr())
        cache = self.__dict__.get("_gdm_root_t")
        if cache is None or cache[0] != key:
            cache = (key, w.detach().t().contiguous())
            self.__dict__["_gdm_root_t"] = cache
        return cache[1]

    def forward_direct_cm(self, x, rowptr, src, attr, relu):
        """First layer (cin <= 16), result CHANNEL-major f32[1, cout, M]: what the next layer's grouped GEMM / root product and the
        final linear read in place (no transposing copy)."""
        return self.forward_direct_cm_packed(x, rowptr, src, attr, relu, False)[0]

    def forward_direct_cm_packed(self, x, rowptr, src, attr, relu, want_packed):
        """... and, want_packed, ALSO as the packed split-bf16 operand of the next layer's grouped GEMM (the kernel writes it itself: no
        pack launch between the layers).  Returns (out_t, packed buffer or None)."""
        M = x.shape[0]
        out_t = torch.empty((1, self.cout, M), dtype=torch.float32, device=x.device)
        pk = ops.spline_packed_buffer(self.cout, M, x.device) if want_packed and self.cout % 128 == 0 and self.cout <= 512 else None
        xc = x.contiguous()
        check(_lib.lib().gdm_spline_direct3_hip(xc.data_ptr(), self.weight.data_ptr(), rowptr.data_ptr(), src.data_ptr(), attr.data_ptr(),
                                                self._root_t().data_ptr(), self.bias.data_ptr(), M, self.cin, self.cout, KERNEL_SIZE, int(relu),
                                                None, out_t.data_ptr(), pk.data_ptr() if pk is not None else None, ops._stream()),
              "gdm_spline_direct3_hip")
        return out_t, pk

    def forward_grouped_cm(self, xt, rowptr, pairs, relu, xpk=None, want_packed=False):
        """128 -> 128 layer on the edge-grouped GEMM, channel-major in (xt f32[1, cin, M]) and out (f32[1, cout, M]).  xpk: xt's packed
        operand when the previous layer wrote it; want_packed: this layer's result also as the next layer's -> (out_t, packed or None)."""
        M = xt.shape[2]
        nk = KERNEL_SIZE ** 3
        wpk, _ = cached_gemm_weight(self, "dense", lambda: self.weight.permute(0, 2, 1).reshape(nk * self.cout, self.cin), (self.weight,))
        Y = ops.gemm_grouped(xt, wpk, pairs["rowidx"], pairs["tile_co0"], nk * self.cout, xpk=xpk)
        root = ops.pointwise([xt], self._root_t(), point_major=True)                       # [1, M, cout] = x @ W_root^T
        out_t = torch.empty((1, self.cout, M), dtype=torch.float32, device=xt.device)
        pk = ops.spline_packed_buffer(self.cout, M, xt.device, avoid=xpk) if want_packed and self.cout % 128 == 0 and self.cout <= 512 else None
        check(_lib.lib().gdm_spline_pairs_aggregate3_hip(Y.data_ptr(), rowptr.data_ptr(), pairs["pos"].data_ptr(), pairs["basis"].data_ptr(),
                                                         root.data_ptr(), self.bias.data_ptr(), M, self.cout, int(relu), None,
                                                         out_t.data_ptr(), pk.data_ptr() if pk is not None else None, ops._stream()),
              "gdm_spline_pairs_aggregate3_hip")
        return out_t, pk

    def forward(self, x, rowptr, src, attr, relu=False, pairs=None):
        M = x.shape[0]
        nk = KERNEL_SIZE ** 3
        if self.cin <= 16 and x.is_cuda and not torch.is_grad_enabled():
            # few input channels (first layer, 9 -> 128): messages formed directly, no [M, 125*out] table (524 MB at M = 8192)
            w = self.lin.weight
            key = (w._version, w.data_ptr())
            cache = self.__dict__.get("_gdm_root_t")
            if cache is None or cache[0] != key:
                cache = (key, w.detach().t().contiguous())
                self.__dict__["_gdm_root_t"] = cache
            out = torch.empty((M, self.cout), dtype=torch.float32, device=x.device)
            xc = x.contiguous()
            check(_lib.lib().gdm_spline_direct_hip(xc.data_ptr(), self.weight.data_ptr(), rowptr.data_ptr(), src.data_ptr(), attr.data_ptr(),
                                                   cache[1].data_ptr(), self.bias.data_ptr(), M, self.cin, self.cout, KERNEL_SIZE, int(relu),
                                                   out.data_ptr(), ops._stream()), "gdm_spline_direct_hip")
            return out
        if (settings.USE_MFMA_GEMM and settings.USE_GROUPED_SPLINE and pairs is not None and not torch.is_grad_enabled() and x.is_cuda
                and self.cin % 128 == 0 and self.cout == 128):
            # edge-grouped form: only the (source, kernel index) pairs some edge needs are multiplied (about a quarter of the dense
            # [M, 125*out] table, which is then never written), on the same split-bf16 MFMA kernel with gathered rows
            wpk, _ = cached_gemm_weight(self, "dense", lambda: self.weight.permute(0, 2, 1).reshape(nk * self.cout, self.cin),
                                        (self.weight,))
            xt = x.t().contiguous().unsqueeze(0)                     # [1, cin, M] channel-major: the GEMM's and the root layer's operand
            Y = ops.gemm_grouped(xt, wpk, pairs["rowidx"], pairs["tile_co0"], nk * self.cout)
            if settings.USE_POINTWISE and self.lin.bias is None:
                w = self.lin.weight
                key = (w._version, w.data_ptr())
                cache = self.__dict__.get("_gdm_root_t")
                if cache is None or cache[0] != key:
                    cache = (key, w.detach().t().contiguous())
                    self.__dict__["_gdm_root_t"] = cache
                root = ops.pointwise([xt], cache[1], point_major=True).view(M, self.cout)      # x @ W_root^T on the own kernel
            else:
                root = self.lin(x)
            out = torch.empty((M, self.cout), dtype=torch.float32, device=x.device)
            check(_lib.lib().gdm_spline_pairs_aggregate_hip(Y.data_ptr(), rowptr.data_ptr(), pairs["pos"].data_ptr(), pairs["basis"].data_ptr(),
                                                            root.data_ptr(), self.bias.data_ptr(), M, self.cout, int(relu), out.data_ptr(),
                                                            ops._stream()), "gdm_spline_pairs_aggregate_hip")
            return out
        if (settings.USE_MFMA_GEMM and not torch.is_grad_enabled() and x.is_cuda
                and ops.gemm_supported(self.cin, nk * self.cout, M)):
            # dense part on the split-bf16 MFMA GEMM, written node-major ([M, 125*out]) as the aggregation kernel reads it
            wpk, _ = cached_gemm_weight(self, "dense", lambda: self.weight.permute(0, 2, 1).reshape(nk * self.cout, self.cin),
                                        (self.weight,))
            xw = ops.gemm_bf16x3(x.t().contiguous().unsqueeze(0), wpk, nk * self.cout, pixel_major=True).view(M, nk, self.cout)
            root = self.lin(x)
            return _SplineAggregate.apply(xw, root, self.bias, rowptr, src, attr, relu)
        w = self.weight.permute(1, 0, 2).reshape(self.cin, -1)            # [in, 125*out]
        xw = torch.matmul(x, w).view(M, KERNEL_SIZE ** 3, self.cout)      # dense GEMM
        root = self.lin(x)
        return _SplineAggregate.apply(xw, root, self.bias, rowptr, src, attr, relu)


def mesh_node_features(model_pts):
    """utils/ply.py:519-535 read_ply_to_data: x = [rgb normalised (std .229,.224,.225), xyz (m), normal]."""
    rgb = model_pts[:, 3:6].astype(np.uint8).astype(np.float64) / 255.0
    rgb = (rgb - COLOR_MEAN.astype(np.float64)) / COLOR_STD_MESH.astype(np.float64)
    xyz = model_pts[:, :3].astype(np.float32) / 1000.0
    x = np.concatenate([rgb, xyz, model_pts[:, 6:9].astype(np.float32)], axis=-1)
    return torch.tensor(x, dtype=torch.float32), torch.tensor(xyz, dtype=torch.float32)


def build_mesh_graph(pos, k=4):
    """KNNGraph(k) + Cartesian() on the GPU. pos f32[M,3] (cuda) -> edge_index i64[2,E] (row 0 = neighbour j,
    row 1 = centre i, grouped by centre, ascending distance), edge_attr f32[E,3]."""
    M = pos.shape[0]
    idx = ops.knn_batch(pos[None], pos[None], k + 1)[0].long()           # [M,k+1], self first (d=0)
    centre = torch.arange(M, device=pos.device).unsqueeze(1).expand(M, k + 1)
    keep = idx != centre                                                 # knn_graph(loop=False): drop self loops
    # exactly one self hit per row unless duplicate vertices tie at distance 0; keep the first k others
    order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)[:, :k]
    nbr = torch.gather(idx, 1, order)
    row = nbr.reshape(-1)
    col = centre[:, :k].reshape(-1)
    cart = pos[row] - pos[col]
    cart = cart / (2 * cart.abs().max()) + 0.5
    return torch.stack([row, col], dim=0), cart


def build_spline_pairs(src, attr, M, cout=128):
    """Static bookkeeping of the edge-grouped SplineConv: the unique (source vertex, kernel index) pairs the edges need, sorted by
    kernel index and padded per kernel index to whole 256-row tiles.  src i32[E] / attr f32[E,3] in CSR (target-sorted) order.
    -> dict(rowidx i32[R], tile_co0 i32[R/256], pos i32[E,8], basis f32[E,8]).  Same fp32 arithmetic as spline_aggregate_kernel."""
    ks = KERNEL_SIZE
    v = attr * float(ks - 1)
    f = torch.floor(v)
    fl = f.to(torch.int64)
    fr = v - f
    E = src.shape[0]
    wi = torch.zeros((E, 8), dtype=torch.int64, device=src.device)
    basis = torch.ones((E, 8), dtype=torch.float32, device=src.device)
    for s_ in range(8):
        off = 1
        for d in range(3):
            kd = (s_ >> d) & 1
            wi[:, s_] += ((fl[:, d] + kd) % ks) * off
            off *= ks
            basis[:, s_] = basis[:, s_] * (fr[:, d] if kd else 1.0 - fr[:, d])
    key = wi * M + src.to(torch.int64)[:, None]                       # sorts by kernel index, then source
    uniq, inv = torch.unique(key.reshape(-1), sorted=True, return_inverse=True)
    pw = uniq // M                                                     # kernel index of every unique pair
    counts = torch.bincount(pw, minlength=ks ** 3)
    padded = (counts + 255) // 256 * 256
    start = torch.cumsum(padded, 0) - padded                          # first row of each kernel index' block
    first = torch.cumsum(counts, 0) - counts                          # first unique-pair id of each kernel index
    row = start[pw] + (torch.arange(uniq.shape[0], device=src.device) - first[pw])
    R = max(int(padded.sum().item()), 256)
    rowidx = torch.zeros(R, dtype=torch.int32, device=src.device)
    rowidx[row] = (uniq % M).to(torch.int32)
    tile_k = torch.repeat_interleave(torch.arange(ks ** 3, device=src.device), padded // 256)
    tile_co0 = torch.zeros(R // 256, dtype=torch.int32, device=src.device)
    tile_co0[: tile_k.shape[0]] = (tile_k * cout).to(torch.int32)
    return dict(rowidx=rowidx.contiguous(), tile_co0=tile_co0.contiguous(), pos=row[inv].view(E, 8).to(torch.int32).contiguous(),
                basis=basis.contiguous())


class SplineCNN_Mesh(nn.Module):
    def __init__(self, cfg, idx, mesh_in_channels=9, out_channels=128, mesh_coord_dim=3, num_mesh_layers=3,
                 cat=True, lin=True, dropout=0.1, model_points=None):
        """cfg keys as the reference (SplineCNN.py:108-110): model_pth, n_mesh_node, model_name.
        `model_points` f32[>=M,9] may be passed instead of reading obj_%06d_fps.npy."""
        super().__init__()
        self.selected_mesh_num = cfg["n_mesh_node"]
        self.name = cfg.get("model_name", "lmo")
        if model_points is None:
            model_points = np.load(os.path.join(cfg["model_pth"], "obj_%06d_fps.npy" % idx))
        model_points = np.asarray(model_points)[: self.selected_mesh_num]
        M = model_points.shape[0]
        x, pos = mesh_node_features(model_points)
        self.register_buffer("xyz", pos)
        self.register_buffer("mesh_graph_x", x)
        self.register_buffer("mesh_graph_edge_index", torch.zeros((2, 4 * M), dtype=torch.int64))
        self.register_buffer("mesh_graph_edge_attr", torch.zeros((4 * M, 3), dtype=torch.float32))
        self.register_buffer("const_one", torch.tensor(1))
        self.graph_ready = False

        self.out_channels = out_channels
        self.cat = cat
        self.dropout = dropout
        self.mesh_convs = nn.ModuleList()
        cin = mesh_in_channels
        for _ in range(num_mesh_layers):
            self.mesh_convs.append(SplineConv(cin, out_channels))
            cin = out_channels
        fin = mesh_in_channels + num_mesh_layers * out_channels if cat else out_channels
        self.mesh_final = nn.Linear(fin, out_channels) if lin else None
        # symmetric objects (SplineCNN.py:155-161): the reference raises NameError there (`misc` is not
        # imported); symmetry correspondences are a "next" row, none for the LineMOD/YCB objects benched.
        self.sys_corr_idx = None
        self._csr = None
        self._pairs = None

    def set_symmetry(self, sys_idx):
        """Symmetric object: sys_idx int[M] = index of each vertex's counterpart under the object's symmetry (the reference's
        `cal_sys_idx`, SplineCNN.py:163-169, registered as buffer `sys_idx` :161).  Switches the training matching loss to
        matching_loss_sys (geoMatch.py:138-141)."""
        idx = torch.as_tensor(sys_idx).long().reshape(-1)
        if idx.shape[0] != self.xyz.shape[0]:
            raise ValueError("sys_idx must have one entry per model vertex (%d), got %d" % (self.xyz.shape[0], idx.shape[0]))
        if "sys_idx" in self._buffers:
            self.sys_idx = idx.to(self.xyz.device)
        else:
            self.register_buffer("sys_idx", idx.to(self.xyz.device))
        self.sys_corr_idx = self.sys_idx

    def set_symmetry_transform(self, R, t_mm):
        """sys_idx from a symmetry transform of the model frame (SplineCNN.py:163-169: `sym_transforms[1]` of
        misc.get_symmetry_transformations, t in mm): for every vertex the nearest vertex of the transformed model, by the HIP kNN."""
        if not self.xyz.is_cuda:
            raise RuntimeError("set_symmetry_transform runs the HIP kNN: move the module to the GPU first")
        Rm = torch.as_tensor(R, dtype=torch.float32, device=self.xyz.device).reshape(3, 3)
        t = torch.as_tensor(t_mm, dtype=torch.float32, device=self.xyz.device).reshape(1, 3) / 1000.0
        moved = (self.xyz @ Rm.t() + t).contiguous()
        self.set_symmetry(ops.knn_batch(moved[None], self.xyz[None].contiguous(), 1)[0, :, 0])

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        ei = state_dict.get(prefix + "mesh_graph_edge_index")
        if ei is not None and bool(torch.as_tensor(ei).any()):
            self.graph_ready = True                                       # a real graph comes with the checkpoint
            self._csr = None
        elif ei is not None:
            # the all-zero placeholder of a state_dict taken before the first GPU forward: keep building the graph here
            state_dict = dict(state_dict)
            state_dict[prefix + "mesh_graph_edge_index"] = self.mesh_graph_edge_index
            if prefix + "mesh_graph_edge_attr" in state_dict:
                state_dict[prefix + "mesh_graph_edge_attr"] = self.mesh_graph_edge_attr
        if prefix + "sys_idx" in state_dict and "sys_idx" not in self._buffers:
            self.set_symmetry(state_dict[prefix + "sys_idx"])
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _ensure_graph(self):
        if not self.graph_ready:
            if not self.xyz.is_cuda:
                raise RuntimeError("SplineCNN_Mesh: the mesh graph is built by the HIP kNN kernel; move the module "
                                   "to the GPU (or load a checkpoint that carries the graph) before calling forward")
            ei, ea = build_mesh_graph(self.xyz, k=4)
            self.mesh_graph_edge_index = ei
            self.mesh_graph_edge_attr = ea
            self.graph_ready = True
            self._csr = None
        if self._csr is None or self._csr[0].device != self.xyz.device:
            ei, ea = self.mesh_graph_edge_index, self.mesh_graph_edge_attr
            M = self.xyz.shape[0]
            order = torch.argsort(ei[1], stable=True)                     # group edges by target
            tgt = ei[1][order]
            rowptr = torch.zeros(M + 1, dtype=torch.int32, device=ei.device)
            rowptr[1:] = torch.cumsum(torch.bincount(tgt, minlength=M), 0).to(torch.int32)
            self._csr = (rowptr.contiguous(), ei[0][order].to(torch.int32).contiguous(), ea[order].contiguous())
            self._pairs = build_spline_pairs(self._csr[1], self._csr[2], M, self.out_channels) if ei.is_cuda and self.out_channels == 128 else None
        return self._csr

    def _channel_major_ok(self):
        convs = list(self.mesh_convs)
        return (settings.USE_POINTWISE and settings.USE_MFMA_GEMM and settings.USE_GROUPED_SPLINE and self.cat and self.mesh_final is not None
                and (not self.training) and not torch.is_grad_enabled() and self.xyz.is_cuda and self._pairs is not None and len(convs) <= 3
                and convs[0].cin <= 16 and convs[0].cout == 128 and all(c.cin % 128 == 0 and c.cout == 128 for c in convs[1:]))

    def _forward_channel_major(self, rowptr, src, attr):
        """Inference: every layer hands its result on channel-major ([1, C, M]), which is what the next layer's grouped GEMM and root
        product read and what the final linear reads as concat-free segments -- no transposing copies, no torch.cat, no library GEMM;
        the result f32[128, M] is the layout GeoMatch.forward returns."""
        x0 = self.mesh_graph_x
        key = (x0._version, x0.data_ptr())
        cache = self.__dict__.get("_gdm_x0_t")
        if cache is None or cache[0] != key:
            cache = (key, x0.detach().t().contiguous().unsqueeze(0))       # [1, 9, M], constant per object
            self.__dict__["_gdm_x0_t"] = cache
        segs = [cache[1]]
        convs = list(self.mesh_convs)
        # each layer's kernel also writes the packed operand of the NEXT layer's grouped GEMM (settings.USE_PACKED_PRODUCERS)
        packed = settings.USE_PACKED_PRODUCERS
        out_t, pk = convs[0].forward_direct_cm_packed(x0, rowptr, src, attr, True, packed and len(convs) > 1)
        segs.append(out_t)
        for li, conv in enumerate(convs[1:]):
            out_t, pk = conv.forward_grouped_cm(segs[-1], rowptr, self._pairs, True, xpk=pk, want_packed=packed and li + 2 < len(convs))
            segs.append(out_t)
        w = self.mesh_final.weight
        key = (w._version, w.data_ptr())
        cache = self.mesh_final.__dict__.get("_gdm_wt")
        if cache is None or cache[0] != key:
            cache = (key, w.detach().t().contiguous())
            self.mesh_final.__dict__["_gdm_wt"] = cache
        out = ops.pointwise(segs, cache[1], None, self.mesh_final.bias)     # [1, 128, M]; eval: dropout is the identity
        return out[0]

    def forward(self):
        rowptr, src, attr = self._ensure_graph()
        if self._channel_major_ok():
            return self._forward_channel_major(rowptr, src, attr)
        feats = [self.mesh_graph_x]
        for conv in self.mesh_convs:
            feats.append(conv(feats[-1], rowptr, src, attr, relu=True, pairs=self._pairs))   # F.relu(conv(...)) (SplineCNN.py:238-239)
        out = torch.cat(feats, dim=-1) if self.cat else feats[-1]
        out = F.dropout(out, p=self.dropout, training=self.training)
        if self.mesh_final is not None:
            out = self.mesh_final(out)
        return out.transpose(0, 1)
