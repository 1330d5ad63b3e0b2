"""GeoMatch: drop-in for /root/reference/models/geoMatch.py (`GeoMatch(cfg, cls_id)`,
`forward(inputs, end_points=None) -> dict`), same sub-module and parameter names
(awl, model_emb, pcd_emb, seg_layer, feature_encoding_layer, normalize_feature_layer), same
end_points keys: seg [B,2,N], mesh [1,128,M], rgbd [B,128,N] and, in training mode, loss /
seg_loss / match_loss (geoMatch.py:159-200).

Inputs are the loader's dict (datasets/lm/linemod_pbr.py:572-599) on the GPU.  Index tensors may be
int32 (as the loader makes them) or int64 (as train_lm.py:167-168 widens them).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops, settings
from .ffb6d import FFB6DEmb
from .layers import PtSeq, pt_conv1d
from .loss import AutomaticWeightedLoss, CircleLoss, FocalLoss
from .splinecnn import SplineCNN_Mesh


def pdist(A, B):
    """utils/basic_utils.py:86-89, 'L2' branch."""
    D2 = torch.sum((A.unsqueeze(1) - B.unsqueeze(0)).pow(2), 2)
    return torch.sqrt(D2 + 1e-7)


class GeoMatch(nn.Module):
    def __init__(self, cfg, cls_id, model_points=None, cache_mesh_in_eval=False):
        super().__init__()
        self.awl = AutomaticWeightedLoss(2)
        self.feat_dim = cfg["feat_dim"]
        self.positive_r = cfg["neighbor_dis_th"] * cfg["model_d"][cls_id] / 1000.0
        self.model_emb = SplineCNN_Mesh(cfg, cls_id, model_points=model_points)
        self.pcd_emb = FFB6DEmb(cfg["ffb_config"])
        self.circle_loss = CircleLoss(16)
        self.ce_loss = nn.CrossEntropyLoss()
        self.seg_loss_func = FocalLoss(gamma=2)

        self.seg_layer = (PtSeq(self.feat_dim).conv1d(128, bn=True).conv1d(128, bn=True).conv1d(128, bn=True)
                          .conv1d(2, activation=None))
        self.feature_encoding_layer = (PtSeq(128).conv1d(128, bn=True).conv1d(128, bn=True).conv1d(128, bn=True)
                                       .conv1d(self.feat_dim, activation=None, bias=False))
        self.normalize_feature_layer = pt_conv1d(self.feat_dim, self.feat_dim, bn=True)

        self.cache_mesh_in_eval = cache_mesh_in_eval
        self._mesh_cache = None

    # ------------------------------------------------------------------ training matching (geoMatch.py:55-157)
    def _positive_tables(self, mesh_xyz):
        """Radius-test bit table of the model (depends on xyz and positive_r only): built once, rebuilt if either changes."""
        from . import ops
        key = (mesh_xyz.data_ptr(), mesh_xyz._version, float(self.positive_r), mesh_xyz.device)
        cache = self.__dict__.get("_gdm_nbr")
        if cache is None or cache[0] != key:
            cache = (key, ops.circle_nbr_table(mesh_xyz, self.positive_r))
            self.__dict__["_gdm_nbr"] = cache
        return cache[1]

    def pointwise_feature_matching(self, rgbd_feature, mesh_feature, x):
        """geoMatch.py:102-157 for the whole batch at once: value and gradients of the reference's per-item loop (mean over the
        items with >= 3 selected points of the mean circle loss of their rows).  The [n_sel, M+1] similarity is never formed: unit
        rows go to ops.circle_match (MFMA similarity tiles, masked LSEs in registers, recomputation in the backward kernels).
        Non-symmetric objects: positives = visible vertices within positive_r of the ground-truth vertex (:55-83); symmetric
        objects (model_emb.sys_corr_idx set): the row's own match and the match of its symmetric counterpart (:86-100, indexing
        restated as the reference writes it).  settings.USE_FUSED_MATCH_LOSS = False: same batch formulation with the similarity
        materialised by one GEMM (the form the fused kernels are tested against, besides oracle/loss_ref.py and the goldens)."""
        from . import ops, settings
        if not rgbd_feature.is_cuda:
            raise RuntimeError("GeoMatch training matching loss runs on the GPU (HIP kernels); there is no CPU fallback")
        B, D, N = rgbd_feature.shape
        mesh = mesh_feature[0]
        M = mesh.shape[1]
        labels = x["labels"]
        sel = labels == 1                                          # [B,N]
        counts = sel.sum(dim=1)
        item_ok = counts >= 3                                      # geoMatch.py:126-127
        sel = sel & item_ok.unsqueeze(1)
        # The selected rows are compacted with one host read of their count; under a hipGraph capture (train_graph.py) nothing
        # may depend on the host, so there every one of the B*N rows is run and the unselected ones get weight zero.
        static_rows = settings.STATIC_MATCH_ROWS or torch.cuda.is_current_stream_capturing()
        if static_rows:
            bi = torch.arange(B, device=mesh.device).repeat_interleave(N)
            pi = torch.arange(N, device=mesh.device).repeat(B)
        else:
            bi, pi = torch.nonzero(sel, as_tuple=True)             # row-major: item, then point order
            if bi.numel() == 0:
                return torch.zeros((), device=mesh.device)
        rows = F.normalize(rgbd_feature.transpose(1, 2)[bi, pi], p=2, dim=1)      # [R,128]
        match_all = x["match_idx"].long()
        if static_rows:
            match_all = match_all.clamp(0, M)                      # M = "no correspondence"; an unselected row's entry may be anything
        symmetric = self.model_emb.sys_corr_idx is not None
        if symmetric:
            if N != M:
                raise IndexError("symmetric matching loss: the reference indexes the per-vertex symmetry table with point indices and "
                                 "the per-point match table with its entries (geoMatch.py:91-93), which needs N == M (got %d, %d)" % (N, M))
            sys_idx = self.model_emb.sys_idx.to(match_all.device).long()
            c1 = match_all[bi, pi]
            c2 = match_all[bi, sys_idx[pi]]
        else:
            c1, c2 = match_all[bi, pi], None
        if settings.USE_FUSED_MATCH_LOSS:
            mesh_rows = F.normalize(mesh, p=2, dim=0).t().contiguous()             # [M,128]; the -1 padding column is analytic
            if symmetric:
                lrow = ops.circle_match(rows, mesh_rows, c1, bi, c2=c2, gamma=16.0, m=0.2)
            else:
                lrow = ops.circle_match(rows, mesh_rows, c1, bi, nbr=self._positive_tables(self.model_emb.xyz.contiguous()),
                                        visb=ops.circle_visbits(x["visible_flag"]), gamma=16.0, m=0.2)
        else:
            padding = -torch.ones((self.feat_dim, 1), dtype=torch.float32, device=mesh.device)
            mesh_padded = F.normalize(torch.cat([mesh, padding], dim=1), p=2, dim=0)
            sim = torch.matmul(rows, mesh_padded)                  # [R, M+1], materialised
            if symmetric:
                mask = torch.zeros(sim.shape, dtype=torch.bool, device=sim.device)
                ar = torch.arange(sim.shape[0], device=sim.device)
                mask[ar, c1] = True
                mask[ar, c2] = True
                lrow = self.circle_loss.rows(sim, mask, 0.2)
            else:
                lrow = ops.circle_rows(sim, c1, bi, self.model_emb.xyz.contiguous(), x["visible_flag"], self.positive_r, 16.0, 0.2)
        if static_rows:
            per_item = (lrow.view(B, N) * sel.to(lrow.dtype)).sum(dim=1) / counts.clamp(min=1).to(torch.float32)
            ok = item_ok.to(torch.float32)
            return (per_item * ok).sum() / ok.sum().clamp(min=1.0)
        per_item = torch.zeros(B, dtype=torch.float32, device=mesh.device).index_add_(0, bi, lrow)
        per_item = per_item[item_ok] / counts[item_ok].to(torch.float32)
        return per_item.mean()

    # ------------------------------------------------------------------ forward (geoMatch.py:159-200)
    def mesh_features(self):
        if self.cache_mesh_in_eval and not self.training:
            if self._mesh_cache is None:
                with torch.no_grad():
                    self._mesh_cache = self.model_emb()
            return self._mesh_cache
        self._mesh_cache = None
        return self.model_emb()

    def _fused_heads(self, rgb):
        """Packed weights + folded BatchNorms of the per-point heads for ops.point_heads, or None when the fused kernel does not apply
        (training / autograd, a non-default head structure, split-bf16 GEMMs switched off).  Cached until a parameter changes."""
        from .layers import act_code, fused_eval, folded_bn, _PtConv
        if not (settings.USE_FUSED_HEADS and settings.USE_MFMA_GEMM and fused_eval(rgb, self)):
            return None
        chain = list(self.feature_encoding_layer) + [self.normalize_feature_layer] + list(self.seg_layer)
        if len(chain) != 9 or not all(isinstance(m, _PtConv) and isinstance(m.conv, nn.Conv1d) for m in chain):
            return None
        deps = []
        for m in chain:
            deps += [m.conv.weight] + ([m.conv.bias] if m.conv.bias is not None else [])
            if hasattr(m, "normlayer"):
                bn = m.normlayer.bn
                deps += [bn.weight, bn.bias, bn.running_mean, bn.running_var]
        key = tuple((t._version, t.data_ptr()) for t in deps)
        cache = self.__dict__.get("_gdm_heads")
        if cache is not None and cache[0] == key:
            return cache[1]
        hidden, last = chain[:-1], chain[-1]
        ok = (all(m.conv.in_channels == 128 and m.conv.out_channels == 128 for m in hidden) and last.conv.in_channels == 128
              and last.conv.out_channels <= 16 and not hasattr(last, "normlayer") and getattr(last, "activation", None) is None)
        codes = [act_code(getattr(m, "activation", None)) for m in hidden]
        ok = ok and all(c is not None and c[0] in (ops.ACT_NONE, ops.ACT_RELU) for c in codes)
        value = None
        if ok:
            with torch.no_grad():
                layers = []
                for m, c in zip(hidden, codes):
                    if hasattr(m, "normlayer"):
                        scale, shift = folded_bn(m.normlayer.bn, m.conv.bias)
                    else:
                        scale, shift = None, (m.conv.bias.detach().contiguous() if m.conv.bias is not None else None)
                    layers.append((ops.gemm_pack_weight(m.conv.weight.reshape(128, 128)), scale, shift, c[0]))
                cl = last.conv.out_channels
                value = (layers, (ops.gemm_pack_weight(last.conv.weight.reshape(cl, 128)),
                                  last.conv.bias.detach().contiguous() if last.conv.bias is not None else None, cl))
        self.__dict__["_gdm_heads"] = (key, value)
        return value

    def forward(self, inputs, end_points=None, defer_seg=False):
        """defer_seg: accepted for callers of earlier rounds (a split of the fused heads onto a side stream was measured without gain and
        removed in round 4); the forward is the same either way."""
        if not end_points:
            end_points = {}
        rgb = inputs["rgb"]
        late_join = None
        mesh_rows = None
        heads = self._fused_heads(rgb)
        # the fused heads read the two halves of the embedding in place: no concat launch
        emb = (lambda x: self.pcd_emb(x, parts=True)) if (heads is not None and isinstance(self.pcd_emb, FFB6DEmb)) else self.pcd_emb
        if settings.USE_SIDE_STREAMS and "mesh" in settings.SIDE_PARTS and (not self.training) and rgb.is_cuda and not torch.is_grad_enabled():
            # the mesh branch depends on nothing in `inputs`: it runs on a side stream beside the RGB-D embedding
            if settings.MESH_FORK_LATE:
                # enqueued BEHIND the embedding (it only waits for an event recorded before it): in a hipGraph the branch is still a
                # root, but the executor -- which spreads a graph over very few hardware queues, in node order -- then keeps the image
                # branch on a queue of its own instead of queueing layer1 behind the mesh kernels (tools/step_sequence.py, Queue_Id)
                ev0 = torch.cuda.Event()
                ev0.record(torch.cuda.current_stream(rgb.device))
                rgbd_emb = emb(inputs)
                with ops.fork(rgb.device, 1, start=ev0) as f:
                    mesh_features = self.mesh_features()
                    # the matching kernel's operand rows of the model descriptors, formed here (off the step's serial tail)
                    mesh_rows = ops.match_pack(mesh_features, ops.MATCH_BF16X3) if settings.PACK_MESH_ROWS else None
                late_join = f if heads is not None else None      # the fused heads do not read the mesh: join behind them
                if late_join is None:
                    f.join(mesh_features, mesh_rows)
            else:
                with ops.fork(rgb.device, 1) as f:           # reads module buffers / parameters only (never freed mid-step)
                    mesh_features = self.mesh_features()
                rgbd_emb = emb(inputs)
                f.join(mesh_features)
        else:
            rgbd_emb = emb(inputs)
            mesh_features = self.mesh_features()
        if heads is not None:
            # feature_encoding_layer, normalize_feature_layer, the residual add and seg_layer: nine per-point 1x1 convolutions, one launch
            a, b = rgbd_emb if isinstance(rgbd_emb, tuple) else (rgbd_emb, None)
            rgbd_features, seg_features = ops.point_heads(a, b, heads[0], heads[1], feat_layer=3, res_layer=4)
            if late_join is not None:
                late_join.join(mesh_features, mesh_rows)
        else:
            rgbd_features = self.feature_encoding_layer(rgbd_emb)
            rgbd_normalized = self.normalize_feature_layer(rgbd_features)
            rgbd_emb = rgbd_emb + rgbd_normalized
            seg_features = self.seg_layer(rgbd_emb)
        mesh_features = mesh_features.unsqueeze(0)

        if self.training:
            match_loss = self.pointwise_feature_matching(rgbd_features, mesh_features, inputs)
            seg_loss = self.seg_loss_func(seg_features, inputs["labels"].long())
            end_points["loss"] = self.awl(seg_loss, match_loss)
            end_points["seg_loss"] = seg_loss
            end_points["match_loss"] = match_loss

        end_points["seg"] = seg_features
        if mesh_rows is not None:
            end_points["mesh_rows"] = mesh_rows          # u8 packed rows for ops.match_packed (matching.match_tail reads them)
        end_points["mesh"] = mesh_features
        end_points["rgbd"] = rgbd_features
        return end_points
