"""GPU parity of the product path (pyramid -> GeoMatch.forward -> matching) against
  (1) golden vectors made by the REAL reference in the build container (tests/golden), and
  (2) the oracle's CPU restatement on the same seeded inputs.
Everything the product computes here runs through libgdm_hip.so + MIOpen; nothing falls back to CPU."""
import json
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, G)
import inputs as gin  # noqa: E402

from geometric_aware_dense_matching_amd import settings, synthetic  # noqa: E402
from geometric_aware_dense_matching_amd.config import make_model_cfg  # noqa: E402


def _dev_inputs(batch):
    d = {k: torch.from_numpy(batch[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose", "labels")}
    d["dpt_xyz"] = torch.from_numpy(batch["dpt_xyz"]).cuda()
    return d


def test_pyramid_bit_exact_vs_reference_golden():
    from geometric_aware_dense_matching_amd import pyramid
    gold = np.load(os.path.join(G, "knn_pyramid_c1.npz"))
    crop = synthetic.make_crop(seed=101, n_points=1024)
    cld = torch.from_numpy(crop["cld_rgb_nrm"][None]).cuda()
    pyr = pyramid.build_pyramid(pyramid.cloud_from_inputs(cld), torch.from_numpy(crop["dpt_xyz"][None]).cuda())
    for key in gold.files:
        if key == "cld_checksum":
            continue
        got = pyr[key][0].cpu().numpy()
        assert got.dtype == np.int32 and np.array_equal(got, gold[key]), key


def test_pyramid_batch_c2_vs_oracle():
    from geometric_aware_dense_matching_amd import pyramid
    from oracle import pyramid as opyr
    B, N = 3, 2048
    batch = synthetic.make_batch(seed=9, batch=B, n_points=N)
    d = _dev_inputs(batch)
    pyr = pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"])
    for b in range(B):
        want = opyr.build_pyramid(batch["cld_rgb_nrm"][b, :3].T.copy(), batch["dpt_xyz"][b])
        for k, v in want.items():
            assert np.array_equal(pyr[k][b].cpu().numpy(), v), (b, k)


def test_pyramid_with_duplicate_points_vs_oracle():
    from geometric_aware_dense_matching_amd import pyramid
    from oracle import pyramid as opyr
    batch = synthetic.make_batch(seed=4, batch=1, n_points=1024, duplicates=True)
    d = _dev_inputs(batch)
    pyr = pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"])
    want = opyr.build_pyramid(batch["cld_rgb_nrm"][0, :3].T.copy(), batch["dpt_xyz"][0])
    for k, v in want.items():
        assert np.array_equal(pyr[k][0].cpu().numpy(), v), k


@pytest.fixture(scope="module")
def golden_model():
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    M = 512
    model = GeoMatch(make_model_cfg(n_mesh_node=M), 1, model_points=synthetic.make_model_points(1, M))
    keys = json.load(open(os.path.join(G, "geomatch_state.json")))
    sd = synthetic.synthetic_state_dict({k: torch.zeros(v) for k, v in keys.items()}, seed=0)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith("model_emb.") for k in missing)
    return model.cuda().eval(), sd


def test_forward_eval_vs_reference_golden(golden_model):
    """Product forward (HIP pyramid + HIP gathers + MIOpen convs) vs the REAL reference's outputs.
    Tolerance 5e-4 x max|ref| absolute (fp32, different conv algorithms, activations up to ~10)."""
    from geometric_aware_dense_matching_amd import pyramid
    model, _ = golden_model
    g = np.load(os.path.join(G, "geomatch_eval.npz"))
    batch = synthetic.make_batch(seed=5, batch=2, n_points=1024)
    d = _dev_inputs(batch)
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))
    with torch.no_grad():
        ep = model(d)
        emb = model.pcd_emb(d)
    assert ep["mesh"].shape == (1, 128, 512) and ep["rgbd"].shape == (2, 128, 1024) and ep["seg"].shape == (2, 2, 1024)
    for name, t in (("emb", emb), ("rgbd", ep["rgbd"]), ("seg", ep["seg"])):
        t = t.float().cpu()
        got = t.reshape(-1)[torch.from_numpy(g[name + "_pos"])].numpy()
        scale = max(1.0, float(np.abs(g[name + "_val"]).max()))
        assert np.abs(got - g[name + "_val"]).max() < 5e-4 * scale, name
        assert abs(t.double().norm().item() - float(g[name + "_norm"])) < 2e-4 * float(g[name + "_norm"])


def test_forward_accepts_int64_indices_like_train_lm(golden_model):
    from geometric_aware_dense_matching_amd import pyramid
    model, _ = golden_model
    batch = synthetic.make_batch(seed=6, batch=1, n_points=1024)
    d = _dev_inputs(batch)
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))
    with torch.no_grad():
        a = model(dict(d))["rgbd"]
        d64 = {k: (v.long() if v.dtype == torch.int32 else v) for k, v in d.items()}
        b = model(d64)["rgbd"]
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-5)        # MIOpen may pick atomically-reduced conv kernels: not bitwise


def test_mesh_branch_vs_oracle(golden_model):
    """SplineCNN branch (parity UNPINNED against the third-party op; self-consistency HIP vs CPU restatement)."""
    from oracle import model_ref
    model, _ = golden_model
    with torch.no_grad():
        got = model.model_emb().cpu()
    sd = {("model_emb." + k): v.cpu() for k, v in model.model_emb.state_dict().items()}
    ei, ea = model_ref.mesh_graph(sd["model_emb.xyz"], k=4)
    assert torch.equal(ei, sd["model_emb.mesh_graph_edge_index"])                  # HIP kNN graph == CPU graph
    assert torch.allclose(ea, sd["model_emb.mesh_graph_edge_attr"], atol=1e-6)
    want = model_ref.spline_mesh_forward(sd)
    assert got.shape == (128, 512)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-4)


def test_mesh_branch_packed_producers_equal_the_pack_launches_bit_for_bit(golden_model):
    """The SplineConv kernels write the next layer's packed split-bf16 operand themselves (gdm_spline_direct3_hip /
    gdm_spline_pairs_aggregate3_hip) == the separate pack launch over their fp32 result: same bytes, so the same mesh descriptors."""
    from geometric_aware_dense_matching_amd import ops
    model, _ = golden_model
    saved = settings.USE_PACKED_PRODUCERS
    try:
        with torch.no_grad():
            settings.USE_PACKED_PRODUCERS = True
            a = model.model_emb().clone()
            settings.USE_PACKED_PRODUCERS = False
            b = model.model_emb().clone()
            # the operand bytes themselves, first layer
            emb = model.model_emb
            rowptr, src, attr = emb._ensure_graph()
            out_t, pk = list(emb.mesh_convs)[0].forward_direct_cm_packed(emb.mesh_graph_x, rowptr, src, attr, True, True)
            mine = pk.clone()
            ref = ops.conv3x3_pack_act(out_t.unsqueeze(2).contiguous())          # [1, C, 1, M]
    finally:
        settings.USE_PACKED_PRODUCERS = saved
    assert torch.equal(a, b)
    assert torch.equal(ref.buf, mine)


def test_training_losses_vs_reference_golden(golden_model):
    model, _ = golden_model
    g = np.load(os.path.join(G, "losses.npz"))
    li = gin.loss_inputs()
    dev = torch.device("cuda")
    saved_xyz, saved_r = model.model_emb.xyz, model.positive_r
    try:
        model.model_emb.xyz = torch.from_numpy(g["mesh_xyz"]).to(dev)
        model.positive_r = float(g["positive_r"])
        rgbd = torch.from_numpy(li["rgbd_f"]).to(dev).requires_grad_(True)
        mesh = torch.from_numpy(li["mesh_f"]).to(dev).requires_grad_(True)
        x = dict(labels=torch.from_numpy(li["labels"]).to(dev), match_idx=torch.from_numpy(li["match_idx"]).to(dev),
                 visible_flag=torch.from_numpy(li["vis"]).to(dev), RT=torch.zeros(2, 3, 4, device=dev))
        ml = model.pointwise_feature_matching(rgbd, mesh, x)
        ml.backward()
        assert abs(ml.item() - float(g["match_loss"])) < 1e-4 * max(1.0, abs(float(g["match_loss"])))
        assert np.allclose(rgbd.grad.cpu().numpy(), g["rgbd_grad"], rtol=1e-3, atol=1e-6)
        assert np.allclose(mesh.grad.cpu().numpy(), g["mesh_grad"], rtol=1e-3, atol=1e-6)
    finally:
        model.model_emb.xyz, model.positive_r = saved_xyz, saved_r
    seg = torch.from_numpy(li["seg"]).to(dev).requires_grad_(True)
    sl = model.seg_loss_func(seg, torch.from_numpy(li["labels"]).to(dev))
    sl.backward()
    assert abs(sl.item() - float(g["seg_loss"])) < 1e-5
    assert np.allclose(seg.grad.cpu().numpy(), g["seg_grad"], rtol=1e-4, atol=1e-7)
    sim = torch.from_numpy(li["sim"]).to(dev).requires_grad_(True)
    cl = model.circle_loss(sim, torch.from_numpy(li["mask"]).to(dev), 0.2)
    cl.backward()
    assert abs(cl.item() - float(g["circle_loss"])) < 1e-4
    assert np.allclose(sim.grad.cpu().numpy(), g["circle_grad"], rtol=1e-3, atol=1e-6)
    with torch.no_grad():
        model.awl.params.copy_(torch.from_numpy(g["awl_params"]))
    tot = model.awl(torch.tensor(float(g["seg_loss"]), device=dev), torch.tensor(float(g["match_loss"]), device=dev))
    assert abs(tot.item() - float(g["awl_total"])) < 1e-4


def test_matching_vs_reference_golden():
    """evaluator.py:79-93 (executed from the reference's own text for the golden) vs the HIP kernels."""
    from geometric_aware_dense_matching_amd import matching
    g = np.load(os.path.join(G, "matching.npz"))
    mi = gin.matching_inputs()
    seg = torch.from_numpy(mi["seg_features"][None]).cuda()
    rgbd = torch.from_numpy(mi["rgbd_features"][None]).cuda()
    mesh = torch.from_numpy(mi["mesh_features"][None]).cuda()
    for prec in ("bf16x3", "f32"):
        res = matching.match_frames(dict(seg=seg, rgbd=rgbd, mesh=mesh), precision=prec)
        mask = res["mask"][0].cpu().numpy().astype(bool)
        assert np.array_equal(mask, g["cls_msk"])
        idx = res["best_idx"][0].cpu().numpy()[mask]
        val = res["best_sim"][0].cpu().numpy()[mask]
        assert np.abs(val - g["max_th"]).max() < 1e-4
        assert (idx == g["obj_pts_idx"]).mean() > 0.999
        sel_idx, sel_val = matching.selected(res, 0)
        assert np.array_equal(sel_idx.cpu().numpy(), idx) and np.array_equal(sel_val.cpu().numpy(), val)


def test_end_to_end_vs_oracle_c1(golden_model):
    """C1-sized crop through the whole path; oracle = CPU restatement driven by the same state_dict."""
    from geometric_aware_dense_matching_amd import matching, pyramid
    from oracle import model_ref, ops_ref
    from oracle import pyramid as opyr
    model, sd = golden_model
    batch = synthetic.make_batch(seed=77, batch=1, n_points=1024)
    d = _dev_inputs(batch)
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))
    with torch.no_grad():
        ep = model(d)
        res = matching.match_frames(ep, precision="bf16x3")
    cpu_in = {k: torch.from_numpy(batch[k]) for k in ("rgb", "cld_rgb_nrm", "choose")}
    cpu_in.update({k: torch.from_numpy(v[None]) for k, v in
                   opyr.build_pyramid(batch["cld_rgb_nrm"][0, :3].T.copy(), batch["dpt_xyz"][0]).items()})
    mesh = ep["mesh"][0].cpu()
    with torch.no_grad():
        want = model_ref.geomatch_forward(sd, cpu_in, mesh)
    assert torch.allclose(ep["rgbd"].cpu(), want["rgbd"], rtol=1e-3, atol=5e-3)
    assert torch.allclose(ep["seg"].cpu(), want["seg"], rtol=1e-3, atol=5e-3)
    # matching of the PRODUCT's descriptors, checked against the oracle's matching lines on the same descriptors
    wv, wi, ws = ops_ref.match_argmax(ep["rgbd"][0].cpu(), mesh)
    assert (res["best_sim"][0].cpu() - wv).abs().max() < 1e-4
    at = ws.gather(1, res["best_idx"][0].cpu().long().unsqueeze(1)).squeeze(1)
    assert ((wv - at) < 1e-4).all()


def test_dgcnn_variant_vs_reference_golden_and_oracle():
    """geoMatch_DGCNN (BASELINE config 4): product on the GPU vs the real reference's outputs.  The dynamic
    graph comes from fp32 GEMM distances, so a few near-tie neighbours may differ (see test_oracle_model)."""
    from geometric_aware_dense_matching_amd.geoMatch_DGCNN import GeoMatch as GeoMatchDGCNN
    g = np.load(os.path.join(G, "dgcnn_eval.npz"))
    keys = json.load(open(os.path.join(G, "dgcnn_state.json")))
    model = GeoMatchDGCNN(dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, n_mesh_node=384), 1,
                          model_points=synthetic.make_model_points(1, 384))
    model.model_emb.k = 20
    sd = synthetic.synthetic_state_dict({k: torch.zeros(v) for k, v in keys.items() if k != "model_emb.mesh"}, seed=9)
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    x = torch.from_numpy(synthetic.make_batch(seed=8, batch=2, n_points=512)["cld_rgb_nrm"]).cuda()
    with torch.no_grad():
        ep = model(dict(cld_rgb_nrm=x))
        emb = model.pcd_emb(x)
    from geometric_aware_dense_matching_amd import dgcnn
    idx3 = dgcnn.knn(x[:, :3].contiguous(), 16).cpu().numpy()
    assert (np.sort(idx3, axis=-1) == np.sort(g["knn_xyz"], axis=-1)).mean() > 0.995
    for name, t in (("emb", emb), ("rgbd", ep["rgbd"]), ("seg", ep["seg"]), ("mesh", ep["mesh"])):
        t = t.float().cpu()
        assert list(t.shape) == list(g[name + "_shape"])
        got = t.reshape(-1)[torch.from_numpy(g[name + "_pos"])].numpy()
        scale = max(1.0, float(np.abs(g[name + "_val"]).max()))
        assert (np.abs(got - g[name + "_val"]) < 5e-4 * scale).mean() > 0.99, name
        assert abs(t.double().norm().item() - float(g[name + "_norm"])) < 2e-3 * float(g[name + "_norm"])
    # training mode runs end to end (losses finite, gradients reach both trunks)
    model.train()
    rs = np.random.RandomState(0)
    inp = dict(cld_rgb_nrm=x, labels=torch.from_numpy((rs.rand(2, 512) < 0.5).astype(np.int64)).cuda(),
               origin_labels=torch.from_numpy((rs.rand(2, 512) < 0.5).astype(np.int64)).cuda(),
               match_idx=torch.from_numpy(rs.randint(0, 385, size=(2, 512)).astype(np.int64)).cuda(),
               visible_flag=torch.from_numpy((rs.rand(2, 384) < 0.6).astype(np.float32)).cuda(),
               RT=torch.tensor([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0.9]], dtype=torch.float32).repeat(2, 1, 1).cuda())
    out = model(inp)
    out["loss"].backward()
    assert torch.isfinite(out["loss"]) and model.pcd_emb.conv1[0].weight.grad.abs().sum() > 0
    assert model.model_emb.conv1[0].weight.grad.abs().sum() > 0


def test_fused_matching_loss_equals_materialised_form_and_oracle(golden_model):
    """ops.circle_match (MFMA similarity tiles + masked LSEs in registers, recomputation in the backward kernels; no [R, M+1]
    tensor) vs (a) the same batch formulation with the similarity materialised (settings.USE_FUSED_MATCH_LOSS = False) and
    (b) the oracle's reference-shaped per-item loop on the CPU: loss and gradients w.r.t. both descriptor sets."""
    from oracle import loss_ref
    model, _ = golden_model
    rs = np.random.RandomState(5)
    B, N, M = 3, 300, 512
    dev = torch.device("cuda")
    x = dict(labels=torch.from_numpy((rs.rand(B, N) < 0.5).astype(np.int64)).to(dev),
             match_idx=torch.from_numpy(rs.randint(0, M + 1, size=(B, N)).astype(np.int64)).to(dev),
             visible_flag=torch.from_numpy((rs.rand(B, M) < 0.5).astype(np.uint8)).to(dev), RT=torch.zeros(B, 3, 4, device=dev))
    x["labels"][2] = 0                                      # an item with < 3 selected points is skipped
    x["labels"][2, :2] = 1
    rgbd0 = torch.from_numpy(rs.randn(B, 128, N).astype(np.float32))
    mesh0 = torch.from_numpy(rs.randn(1, 128, M).astype(np.float32))
    saved = model.positive_r
    model.positive_r = 0.02
    try:
        outs = []
        for fused in (True, False):
            settings.USE_FUSED_MATCH_LOSS = fused
            rgbd = rgbd0.clone().to(dev).requires_grad_(True)
            mesh = mesh0.clone().to(dev).requires_grad_(True)
            loss = model.pointwise_feature_matching(rgbd, mesh, x)
            loss.backward()
            outs.append((loss.item(), rgbd.grad.cpu(), mesh.grad.cpu()))
        rg = rgbd0.clone().requires_grad_(True)
        mg = mesh0.clone().requires_grad_(True)
        want = loss_ref.pointwise_feature_matching(rg, mg, x["labels"].cpu(), x["match_idx"].cpu(), x["visible_flag"].cpu(),
                                                   model.model_emb.xyz.cpu(), 0.02)
        want.backward()
    finally:
        settings.USE_FUSED_MATCH_LOSS, model.positive_r = True, saved
    for got in outs:
        assert abs(got[0] - want.item()) < 2e-5 * max(1.0, abs(want.item()))
        assert torch.allclose(got[1], rg.grad, rtol=2e-3, atol=2e-7)
        assert torch.allclose(got[2], mg.grad, rtol=2e-3, atol=2e-7)


def test_symmetric_object_matching_loss_vs_reference_golden(golden_model):
    """matching_loss_sys (geoMatch.py:86-100,138-141) on the fused kernels: value and gradients produced by the imported
    reference (tests/golden/losses_sym.npz), fused and materialised forms."""
    model, _ = golden_model
    g = np.load(os.path.join(G, "losses_sym.npz"))
    ls = gin.sym_loss_inputs()
    dev = torch.device("cuda")
    x = dict(labels=torch.from_numpy(ls["labels"]).to(dev), match_idx=torch.from_numpy(ls["match_idx"]).to(dev),
             visible_flag=torch.from_numpy(ls["vis"]).to(dev), RT=torch.zeros(ls["labels"].shape[0], 3, 4, device=dev))
    assert model.model_emb.sys_corr_idx is None
    model.model_emb.set_symmetry(ls["sys_idx"])
    try:
        for fused in (True, False):
            settings.USE_FUSED_MATCH_LOSS = fused
            rgbd = torch.from_numpy(ls["rgbd_f"]).to(dev).requires_grad_(True)
            mesh = torch.from_numpy(ls["mesh_f"]).to(dev).requires_grad_(True)
            ml = model.pointwise_feature_matching(rgbd, mesh, x)
            ml.backward()
            assert abs(ml.item() - float(g["match_loss"])) < 1e-4 * max(1.0, abs(float(g["match_loss"]))), fused
            assert np.allclose(rgbd.grad.cpu().numpy(), g["rgbd_grad"], rtol=2e-3, atol=1e-6), fused
            assert np.allclose(mesh.grad.cpu().numpy(), g["mesh_grad"], rtol=2e-3, atol=1e-6), fused
    finally:
        settings.USE_FUSED_MATCH_LOSS = True
        model.model_emb.sys_corr_idx = None
        del model.model_emb._buffers["sys_idx"]


def test_dgcnn_variant_matching_loss_on_fused_kernels_vs_reference_golden():
    """The geoMatch_DGCNN variant's training matching loss (geoMatch_DGCNN.py:52-135) on the fused circle-match kernels -- padding
    column e0, one positive table per item from the per-vertex radius positive_r / 1000 * z(RT v) -- without any [B, N, M + 1] tensor:
    (1) value and both gradients vs the imported reference (tests/golden/dgcnn_losses.npz); (2) at the reference's training shape
    (N = M = 4096, batch 3) vs oracle/loss_ref's restatement of the reference loop."""
    from geometric_aware_dense_matching_amd.geoMatch_DGCNN import GeoMatch as GeoMatchDGCNN
    from oracle import loss_ref
    g = np.load(os.path.join(G, "dgcnn_losses.npz"))
    li = gin.dgcnn_loss_inputs()
    dev = torch.device("cuda")
    Md = g["mesh_xyz"].shape[0]
    dg = GeoMatchDGCNN(dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, n_mesh_node=Md), 1,
                       model_points=synthetic.make_model_points(1, Md)).to(dev).train()
    assert not hasattr(dg, "matching_loss")                                        # the per-item transcription is gone
    with torch.no_grad():
        dg.model_emb.mesh.copy_(torch.from_numpy(g["mesh_buffer"]).to(dev))
    dg.positive_r = float(g["positive_r"])
    rgbd = torch.from_numpy(li["rgbd_f"]).to(dev).requires_grad_(True)
    mesh = torch.from_numpy(li["mesh_f"]).to(dev).requires_grad_(True)
    x = dict(origin_labels=torch.from_numpy(li["origin_labels"]).to(dev), match_idx=torch.from_numpy(li["match_idx"]).to(dev),
             visible_flag=torch.from_numpy(li["vis"]).to(dev), RT=torch.from_numpy(li["RT"]).to(dev))
    torch.cuda.reset_peak_memory_stats()
    ml = dg.pointwise_feature_matching(rgbd, mesh, x)
    ml.backward()
    assert abs(ml.item() - float(g["match_loss"])) < 1e-4 * max(1.0, abs(float(g["match_loss"])))
    assert np.allclose(rgbd.grad.cpu().numpy(), g["rgbd_grad"], rtol=2e-3, atol=1e-6)
    assert np.allclose(mesh.grad.cpu().numpy(), g["mesh_grad"], rtol=2e-3, atol=1e-6)
    # (2) training shape against the oracle loop
    rs = np.random.RandomState(5)
    B, N, M = 3, 4096, 4096
    big = GeoMatchDGCNN(dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, n_mesh_node=M), 1,
                        model_points=synthetic.make_model_points(1, M)).to(dev).train()
    big.positive_r = 12
    mesh_xyz = big.model_emb.mesh[0][:3, :].transpose(0, 1).contiguous()
    RT = np.zeros((B, 3, 4), np.float32)
    for b in range(B):
        q, _ = np.linalg.qr(rs.randn(3, 3))
        RT[b, :, :3] = q if np.linalg.det(q) > 0 else -q
        RT[b, :, 3] = (0.0, 0.0, 0.7 + 0.1 * b)
    xs = dict(origin_labels=torch.from_numpy((rs.rand(B, N) < 0.5).astype(np.int64)), match_idx=torch.from_numpy(rs.randint(0, M + 1, (B, N))),
              visible_flag=torch.from_numpy((rs.rand(B, M) < 0.6).astype(np.float32)), RT=torch.from_numpy(RT))
    rg = torch.from_numpy(rs.randn(B, 128, N).astype(np.float32)).requires_grad_(True)
    mf = torch.from_numpy(rs.randn(1, 128, M).astype(np.float32)).requires_grad_(True)
    want = loss_ref.dgcnn_pointwise_feature_matching(rg, mf, xs["origin_labels"], xs["match_idx"], xs["visible_flag"], xs["RT"],
                                                     mesh_xyz.cpu(), positive_r=12)
    want.backward()
    rgd = rg.detach().to(dev).requires_grad_(True)
    mfd = mf.detach().to(dev).requires_grad_(True)
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    got = big.pointwise_feature_matching(rgd, mfd, {k: v.to(dev) for k, v in xs.items()})
    got.backward()
    torch.cuda.synchronize()
    assert torch.cuda.max_memory_allocated() - base < B * N * (M + 1) * 4 // 2       # no [B, N, M + 1] similarity (201 MB) was ever formed
    assert abs(got.item() - want.item()) < 2e-5 * max(1.0, abs(want.item()))
    assert torch.allclose(rgd.grad.cpu(), rg.grad, rtol=2e-3, atol=2e-7)
    assert torch.allclose(mfd.grad.cpu(), mf.grad, rtol=2e-3, atol=2e-7)


def test_fused_matching_loss_training_shape_and_empty_positive_sets():
    """Default training shape of the reference (N = M = 4096, config/lmo_cfg.py:95-98) on 2 items: fused == materialised form;
    rows whose ground-truth vertex is invisible (empty positive set) give loss 0 and gradient 0, no NaN."""
    from geometric_aware_dense_matching_amd import ops
    rs = np.random.RandomState(9)
    dev = torch.device("cuda")
    R, M, B = 3000, 4096, 2
    xyz = torch.from_numpy((rs.rand(M, 3).astype(np.float32) - 0.5) * 0.1).to(dev)
    vis = torch.from_numpy((rs.rand(B, M) < 0.6).astype(np.uint8)).to(dev)
    g = torch.from_numpy(rs.randint(0, M + 1, size=R).astype(np.int32)).to(dev)
    item = torch.from_numpy(np.sort(rs.randint(0, B, size=R)).astype(np.int32)).to(dev)
    x0 = torch.nn.functional.normalize(torch.from_numpy(rs.randn(R, 128).astype(np.float32)), dim=1).to(dev)
    y0 = torch.nn.functional.normalize(torch.from_numpy(rs.randn(M, 128).astype(np.float32)), dim=1).to(dev)
    radius = 0.006
    nbr, visb = ops.circle_nbr_table(xyz, radius), ops.circle_visbits(vis)
    w = torch.from_numpy(rs.rand(R).astype(np.float32)).to(dev)
    xa, ya = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
    la = ops.circle_match(xa, ya, g, item, nbr=nbr, visb=visb)
    (la * w).sum().backward()
    xb, yb = x0.clone().requires_grad_(True), y0.clone().requires_grad_(True)
    pad = torch.full((1, 128), -1.0 / np.sqrt(128.0), device=dev)
    sim = xb @ torch.cat([yb, pad], dim=0).t()
    lb = ops.circle_rows(sim, g, item, xyz, vis, radius)
    (lb * w).sum().backward()
    assert torch.isfinite(la).all() and torch.isfinite(xa.grad).all() and torch.isfinite(ya.grad).all()
    assert torch.allclose(la, lb, rtol=1e-4, atol=1e-5)
    # gradients are sums over 4097 / 3000 terms of split-bf16 products (2^-17 relative each): compare against the tensor's scale
    ex = (xa.grad - xb.grad).abs().max().item() / xb.grad.abs().max().item()
    ey = (ya.grad - yb.grad).abs().max().item() / yb.grad.abs().max().item()
    print("fused matching loss, N = M = 4096 shape: max gradient error / max gradient: x %.2e, y %.2e" % (ex, ey))
    assert ex < 1e-4 and ey < 1e-4, (ex, ey)
    empty = (g.long() < M) & (vis[item.long(), g.long().clamp(max=M - 1)] == 0)
    # the ground-truth vertex itself is the nearest positive candidate: invisible and no other visible vertex within the radius
    lonely = empty & (la == 0)
    assert int(lonely.sum()) > 0 and bool((xa.grad[lonely] == 0).all())


def test_gpu_pose_solve_vs_reference_golden():
    """Batched masked Kabsch + ADD/ADI on the device vs best_fit_transform / add / adi of the reference
    (executed from its own source text for the golden).  fp64 statistics: poses agree to 1e-6."""
    from geometric_aware_dense_matching_amd import pose
    g = np.load(os.path.join(G, "pose.npz"))
    pi = gin.pose_inputs()
    res = dict(mask=torch.from_numpy(pi["mask"]).cuda(), best_idx=torch.from_numpy(pi["idx"]).cuda())
    model = torch.from_numpy(pi["model"]).cuda()
    RT, valid = pose.solve_poses(res, torch.from_numpy(pi["cld"]).cuda(), model)
    assert valid.cpu().tolist() == [True, True, False]
    assert np.abs(RT.cpu().numpy() - g["RT"]).max() < 2e-6
    gt = torch.from_numpy(pi["RT"]).cuda()
    add = pose.add_metric(RT[:2], gt[:2], model).cpu().numpy()
    adi = pose.adi_metric(RT[:2], gt[:2], model).cpu().numpy()
    assert np.abs(add - g["add"][:2]).max() < 1e-6 and np.abs(adi - g["adi"][:2]).max() < 1e-6
    det = torch.linalg.det(RT[:2, :, :3].double())
    assert torch.allclose(det, torch.ones_like(det), atol=1e-6)


def test_fused_eval_path_equals_module_path(golden_model):
    """Eval under no_grad uses the fused BN+activation(+residual) kernel; with autograd enabled the torch modules run.
    Both must give the same forward (1e-5 relative: BN folded into scale/shift changes rounding)."""
    from geometric_aware_dense_matching_amd import pyramid
    model, _ = golden_model
    batch = synthetic.make_batch(seed=31, batch=2, n_points=1024)
    d = _dev_inputs(batch)
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))
    with torch.no_grad():
        a = model(dict(d))
    b = model(dict(d))                                   # grad enabled -> module path
    for k in ("rgbd", "seg"):
        assert torch.allclose(a[k], b[k].detach(), rtol=1e-4, atol=2e-4), k


def test_multi_object_driver_equals_per_instance_forwards():
    """infer.run_multi_object (instances grouped per object, batched) vs the reference's way (one batch-1 forward per
    instance through that instance's model, train_lm.py:298-314)."""
    from geometric_aware_dense_matching_amd import infer, matching, pyramid
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    models = {}
    for cid in (1, 5):
        m = GeoMatch(make_model_cfg(n_mesh_node=256), cid, model_points=synthetic.make_model_points(cid, 256, 150.0))
        sd = synthetic.synthetic_state_dict({k: v for k, v in m.state_dict().items()
                                             if not k.startswith("model_emb.mesh_graph") and k not in ("model_emb.xyz", "model_emb.const_one")}, seed=cid)
        m.load_state_dict(sd, strict=False)
        models[cid] = m.cuda().eval()
    cls = [5, 1, 5, 1, 1]
    batch = synthetic.make_batch(seed=55, batch=len(cls), n_points=1024)
    d = _dev_inputs(batch)
    got = infer.run_multi_object(models, d, cls)
    assert got["rgbd"].shape == (5, 128, 1024) and got["mesh"].shape == (5, 128, 256) and got["RT"].shape == (5, 3, 4)
    for i, cid in enumerate(cls):
        one = {k: v[i:i + 1] for k, v in d.items()}
        one.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(one["cld_rgb_nrm"]), one["dpt_xyz"]))
        with torch.no_grad():
            ep = models[cid](one)
            res = matching.match_frames(ep)
        assert torch.allclose(got["rgbd"][i], ep["rgbd"][0], rtol=1e-4, atol=1e-4)
        assert torch.allclose(got["seg"][i], ep["seg"][0], rtol=1e-4, atol=1e-4)
        assert torch.equal(got["mesh"][i], ep["mesh"][0])
        assert (got["best_idx"][i] == res["best_idx"][0]).float().mean().item() > 0.995


def test_gpu_front_end_vs_synthetic_generator(golden_model):
    """frontend.make_inputs (depth -> xyz crop -> sampling -> pyramid on the device) against the host generator's
    arithmetic (same dpt_2_pcld formula) and the oracle pyramid; then a forward runs on it."""
    from geometric_aware_dense_matching_amd import frontend
    from oracle import pyramid as opyr
    model, _ = golden_model
    rs = np.random.RandomState(77)
    depth, rgb, nrm = synthetic.make_frame(rs)
    S, N = 256, 1024
    y0, x0 = (480 - S) // 2, (640 - S) // 2
    want_xyz = synthetic.depth_to_xyz(depth)[y0:y0 + S, x0:x0 + S]
    dev = torch.device("cuda")
    rgb_n = torch.from_numpy(synthetic.normalize_color(rgb).transpose(2, 0, 1)[None].copy()).to(dev)
    inp = frontend.make_inputs(rgb_n, torch.from_numpy(depth[None]).to(dev), torch.from_numpy(nrm.transpose(2, 0, 1)[None].copy()).to(dev),
                               torch.from_numpy(synthetic.LM_K[None]).to(dev), torch.tensor([[x0, y0]], dtype=torch.int32, device=dev), S, N)
    assert np.allclose(inp["dpt_xyz"][0].cpu().numpy(), want_xyz, rtol=1e-6, atol=1e-7)
    ch = inp["choose"][0, 0].cpu().numpy()
    assert len(np.unique(ch)) == N and (want_xyz.reshape(-1, 3)[ch, 2] > 1e-6).all()           # valid, without replacement
    cld = inp["cld_rgb_nrm"][0, :3].t().cpu().numpy()
    assert np.array_equal(cld, inp["dpt_xyz"][0].reshape(-1, 3)[torch.from_numpy(ch).long().cuda()].cpu().numpy())
    want = opyr.build_pyramid(cld.copy(), inp["dpt_xyz"][0].cpu().numpy())
    for k, v in want.items():
        assert np.array_equal(inp[k][0].cpu().numpy(), v), k
    with torch.no_grad():
        ep = model(inp)
    assert torch.isfinite(ep["rgbd"]).all() and ep["rgbd"].shape == (1, 128, N)


@pytest.mark.parametrize("forked", [False, True])
def test_graphed_pipeline_equals_eager(golden_model, forked):
    """One hipGraph replay of pyramid + forward + matching + pose == the eager step on new inputs (bit-exact: the eval step has no float atomics).
    forked: captured with settings.USE_SIDE_STREAMS (two-stream pipeline, mesh branch and pyramid on their own streams = parallel branches
    of the graph) and replayed five times back to back; the reference is the single-stream eager step either way."""
    from geometric_aware_dense_matching_amd import infer, matching, pose, pyramid, settings
    model, _ = golden_model
    b0 = _dev_inputs(synthetic.make_batch(seed=61, batch=2, n_points=1024))
    b0.pop("labels")
    saved = settings.USE_SIDE_STREAMS
    try:
        settings.USE_SIDE_STREAMS = forked
        gp = infer.GraphedPipeline(model, b0)
        b1 = _dev_inputs(synthetic.make_batch(seed=62, batch=2, n_points=1024))
        b1.pop("labels")
        if forked:
            for _ in range(4):
                gp(b1)
        got = {k: v.clone() for k, v in gp(b1).items()}
    finally:
        settings.USE_SIDE_STREAMS = saved
    d = dict(b1)
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))
    with torch.no_grad():
        ep = model(d)
        res = matching.match_frames(ep)
        RT, valid = pose.solve_poses(res, d["cld_rgb_nrm"], model.model_emb.xyz)
    # same shapes, same kernels, no float atomics in the eval step: a replay on new inputs equals the eager step bit for bit
    for k in ("rgbd", "seg", "mesh"):
        assert torch.equal(got[k], ep[k]), k
    for k in ("mask", "best_idx", "best_sim"):
        assert torch.equal(got[k], res[k]), k
    assert torch.equal(got["valid"], valid)
    if bool(valid.all()):
        assert torch.allclose(got["RT"], RT, atol=1e-3)


def test_spline_direct_first_layer_equals_dense_form():
    """SplineConv with few input channels: direct message kernel == dense GEMM + CSR aggregation (the training-path form)."""
    from geometric_aware_dense_matching_amd import splinecnn
    torch.manual_seed(3)
    M = 1500
    pos = torch.rand(M, 3, device="cuda")
    ei, ea = splinecnn.build_mesh_graph(pos, k=4)
    order = torch.argsort(ei[1], stable=True)
    rowptr = torch.zeros(M + 1, dtype=torch.int32, device="cuda")
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1][order], minlength=M), 0).to(torch.int32)
    src, attr = ei[0][order].to(torch.int32).contiguous(), ea[order].contiguous()
    conv = splinecnn.SplineConv(9, 128).cuda()
    conv.bias.data.normal_(0, 0.1)
    x = torch.randn(M, 9, device="cuda")
    with torch.enable_grad():
        dense = conv(x, rowptr, src, attr, relu=True).detach()
    with torch.no_grad():
        direct = conv(x, rowptr, src, attr, relu=True)
    assert (dense - direct).abs().max().item() < 1e-5 * max(1.0, dense.abs().max().item())


def test_spline_grouped_form_equals_dense_form():
    """SplineConv(128,128): edge-grouped gathered GEMM + pair aggregation == dense [M,125*128] GEMM + CSR aggregation (both on the
    split-bf16 kernel: same products, differences only in fp32 summation order of the 8 x deg terms)."""
    from geometric_aware_dense_matching_amd import splinecnn
    torch.manual_seed(5)
    M = 2048
    pos = torch.rand(M, 3, device="cuda")
    ei, ea = splinecnn.build_mesh_graph(pos, k=4)
    order = torch.argsort(ei[1], stable=True)
    rowptr = torch.zeros(M + 1, dtype=torch.int32, device="cuda")
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1][order], minlength=M), 0).to(torch.int32)
    src, attr = ei[0][order].to(torch.int32).contiguous(), ea[order].contiguous()
    pairs = splinecnn.build_spline_pairs(src, attr, M)
    conv = splinecnn.SplineConv(128, 128).cuda()
    conv.bias.data.normal_(0, 0.1)
    x = torch.randn(M, 128, device="cuda")
    with torch.no_grad():
        dense = conv(x, rowptr, src, attr, relu=True)
        grouped = conv(x, rowptr, src, attr, relu=True, pairs=pairs)
    assert (dense - grouped).abs().max().item() < 1e-5 * max(1.0, dense.abs().max().item())
    assert pairs["rowidx"].shape[0] < M * 125 // 2          # well under the dense table's row count


@pytest.mark.parametrize("B,H,W", [(2, 32, 32), (1, 19, 27), (3, 8, 40)])
def test_fused_upconv64_equals_upsample_conv_bn_prelu(B, H, W):
    """PSPUpsample(64,64) in eval: the one-kernel form (MFMA channel mix in LDS + 9-tap gather) == Upsample + Conv3x3 + BN + PReLU in
    fp64 torch, and == the two-kernel form."""
    from geometric_aware_dense_matching_amd import cnn
    torch.manual_seed(B * 7 + H)
    mod = cnn.PSPUpsample(64, 64).cuda().eval()
    bn = mod.conv[2]
    bn.running_mean.normal_(0, 0.3); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.2)
    x = torch.randn(B, 64, H, W, device="cuda")
    with torch.no_grad():
        settings.USE_FUSED_UPCONV = True
        got = mod(x)
        settings.USE_FUSED_UPCONV = False
        two = mod(x)
        settings.USE_FUSED_UPCONV = True
        conv, act = mod.conv[1], mod.conv[3]
        up = torch.nn.functional.interpolate(x.double(), size=(2 * H, 2 * W), mode="bilinear", align_corners=True)
        y = torch.nn.functional.conv2d(up, conv.weight.double(), conv.bias.double(), padding=1)
        y = (y - bn.running_mean.double()[None, :, None, None]) / torch.sqrt(bn.running_var.double() + bn.eps)[None, :, None, None]
        y = y * bn.weight.double()[None, :, None, None] + bn.bias.double()[None, :, None, None]
        ref = torch.where(y > 0, y, y * act.weight.double()).float()
    scale = max(1.0, ref.abs().max().item())
    assert (got - ref).abs().max().item() < 3e-5 * scale
    assert (got - two).abs().max().item() < 3e-5 * scale


def test_every_runtime_switch_off_gives_the_same_eval_forward(golden_model):
    """settings.py: each A/B switch selects between a HIP path and the plain form it replaces; flipping any ONE of them off (and
    all of them off together) must reproduce the default eval forward (2e-4 x scale: split-bf16 products vs fp32 libraries)."""
    from geometric_aware_dense_matching_amd import pyramid
    model, _ = golden_model
    batch = synthetic.make_batch(seed=41, batch=2, n_points=1024)
    d = _dev_inputs(batch)
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"]))

    def run():
        with torch.no_grad():
            ep = model(dict(d))
        return ep["rgbd"].clone(), ep["seg"].clone(), ep["mesh"].clone()
    base = run()
    saved = {n: getattr(settings, n) for n in settings.ALL_SWITCHES}
    try:
        for case in [(n,) for n in settings.ALL_SWITCHES] + [tuple(settings.ALL_SWITCHES)]:
            for n in settings.ALL_SWITCHES:
                setattr(settings, n, n not in case)
            got = run()
            for a, b, name in zip(got, base, ("rgbd", "seg", "mesh")):
                scale = max(1.0, b.abs().max().item())
                assert (a - b).abs().max().item() < 2e-4 * scale, (case, name)
    finally:
        for n, v in saved.items():
            setattr(settings, n, v)
