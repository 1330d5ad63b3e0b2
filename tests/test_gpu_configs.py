"""GPU parity at the workloads of BASELINE configs 4 and 5 (configs 1 and 2 are tests/test_gpu_model.py and
tests/test_gpu_headline.py; config 3's per-GPU step is tests/test_gpu_train.py, its two-rank DDP form tests/test_distributed_cpu.py
and tools/ddp_rehearsal.py):

  config 4  geoMatch_DGCNN variant at N = 2048 scene points x M = 8192 model vertices (batch 2) against the oracle's CPU
            restatement (pinned by the reference-made dgcnn_eval.npz at a small size): all entries with the oracle's dynamic
            graphs injected, and the product's own graphs equal to the oracle's except at fp32 near-ties
  config 5  LM-O evaluation pipeline: one model per LM-O object (8), instances of a batch dispatched per object, dense matching,
            batched Kabsch pose and ADD / ADI -- against the reference's lines restated in oracle/ (evaluator.py:78-100 matching +
            best_fit_transform + pysixd add / adi, pinned by matching.npz / pose.npz) applied to the same descriptors
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from geometric_aware_dense_matching_amd import synthetic  # noqa: E402


def test_config4_dgcnn_variant_at_headline_shape_vs_oracle():
    from geometric_aware_dense_matching_amd import dgcnn
    from geometric_aware_dense_matching_amd.config import make_dgcnn_cfg
    from geometric_aware_dense_matching_amd.geoMatch_DGCNN import GeoMatch as GeoMatchDGCNN
    from oracle import dgcnn_ref, model_ref
    N, M, B = 2048, 8192, 2
    torch.manual_seed(0)
    model = GeoMatchDGCNN(make_dgcnn_cfg(n_mesh_node=M, dataset="ycbv"), 2, model_points=synthetic.make_model_points(2, M, 269.573))
    sd = synthetic.synthetic_state_dict({k: v for k, v in model.state_dict().items() if k != "model_emb.mesh"}, seed=4)
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    sd_cpu = {k: v.cpu() for k, v in model.state_dict().items()}
    x = torch.from_numpy(synthetic.make_batch(seed=44, batch=B, n_points=N)["cld_rgb_nrm"])
    torch.set_num_threads(16)
    with torch.no_grad():
        km = model.model_emb.k                                      # 16: the cfg's `k` wins over DgcnnMeshEmb's default of 20 (dgcnn.py:143)
        want = dgcnn_ref.geomatch_dgcnn_forward(sd_cpu, x, k_cloud=16, k_mesh=km)
        graphs = (dgcnn_ref.trunk_graphs(x, model_ref.SD(sd_cpu, "pcd_emb."), 16)
                  + dgcnn_ref.trunk_graphs(sd_cpu["model_emb.mesh"], model_ref.SD(sd_cpu, "model_emb."), km))
    # (1) arithmetic: the oracle's six graphs injected -> every entry within tolerance
    real_knn = dgcnn.knn
    queue = [g[0].to(torch.int32).cuda() for g in graphs]
    dgcnn.knn = lambda feat, k: queue.pop(0)
    try:
        with torch.no_grad():
            ep = model(dict(cld_rgb_nrm=x.cuda()))
    finally:
        dgcnn.knn = real_knn
    assert not queue
    assert ep["rgbd"].shape == (B, 128, N) and ep["mesh"].shape == (1, 128, M) and ep["seg"].shape == (B, 2, N)
    for name in ("rgbd", "seg", "mesh"):
        a, b = ep[name].cpu(), want[name]
        assert (a - b).abs().max().item() < 5e-4 * max(1.0, b.abs().max().item()), name
    # (2) the product's own dynamic graphs (fp32 GEMM + HIP top-k) == the oracle's, up to near-ties at the k-th place
    seen = []

    def recording_knn(feat, k):
        idx = real_knn(feat, k)
        seen.append(idx.long().cpu())
        return idx
    dgcnn.knn = recording_knn
    try:
        with torch.no_grad():
            ep2 = model(dict(cld_rgb_nrm=x.cuda()))
    finally:
        dgcnn.knn = real_knn
    assert len(seen) == 6
    for (want_idx, dist), idx in zip(graphs[:1] + graphs[3:4], seen[:1] + seen[3:4]):     # xyz graphs: same inputs on both sides
        same = (torch.sort(idx, -1)[0] == torch.sort(want_idx, -1)[0]).all(dim=-1).float().mean().item()
        assert same > 0.995
        assert dgcnn_ref.graph_mismatch_not_near_tie(idx, want_idx, dist, tol=2e-4) == 0
    for name in ("rgbd", "seg", "mesh"):                            # free-running: feature-space graphs may flip near-ties
        a, b = ep2[name].cpu().double(), want[name].double()
        assert abs(a.norm().item() - b.norm().item()) < 2e-3 * b.norm().item(), name


def test_config5_lmo_eval_pipeline_pose_and_add_vs_reference_lines():
    from geometric_aware_dense_matching_amd import config, infer, pose, train_lm
    from oracle import ops_ref, pose_ref
    N, M = 4096, 4096                                              # config/lmo_cfg.py:95-98
    ds = config.dataset_config("lmo")
    args = train_lm.build_parser().parse_args(("-state=test --n-points %d --n-mesh %d" % (N, M)).split())
    ids = sorted(ds["objs"])
    assert len(ids) == 8
    torch.manual_seed(1)
    models = {cid: train_lm.build_model(args, cid, cache_mesh_in_eval=True).cuda().eval() for cid in ids}
    cls = [ids[i % 8] for i in range(12)]                          # 12 detected instances over the 8 objects, interleaved
    batch = synthetic.make_batch(seed=71, batch=len(cls), n_points=N)
    d = {k: torch.from_numpy(batch[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
    out = infer.run_multi_object(models, d, cls)
    assert out["RT"].shape == (12, 3, 4) and out["best_idx"].shape == (12, N) and out["mesh"].shape == (12, 128, M)
    rs = np.random.RandomState(3)
    # (A) evaluator.py:79-93 on the product's descriptors: selected-point mask, arg-max vertex and similarity per instance
    for i, cid in enumerate(cls):
        msk = ops_ref.seg_mask(out["seg"][i].cpu())
        assert torch.equal(msk, out["mask"][i].cpu().bool())
        if int(msk.sum()) < 5:
            assert not bool(out["valid"][i])
            continue
        val, idx, _ = ops_ref.match_argmax(out["rgbd"][i].cpu(), out["mesh"][i].cpu(), msk)
        got_idx = out["best_idx"][i].cpu()[msk].long()
        assert (out["best_sim"][i].cpu()[msk] - val).abs().max().item() < 1e-4
        assert (got_idx == idx).float().mean().item() > 0.999
    # (B) evaluator.py:94-100 + pysixd add / adi on WELL-POSED correspondences of the same 12 instances (an untrained network maps
    # every point to nearly the same vertex, which leaves the rotation undetermined): each instance's points are its object's
    # vertices under a known pose + 2 mm noise + 10 % outliers; the solve runs per object group exactly as in the driver
    RTs, adds, adis = [], [], []
    for i, cid in enumerate(cls):
        model_xyz = models[cid].model_emb.xyz.cpu().numpy()
        idx = rs.randint(0, M, size=N).astype(np.int32)
        mask = (rs.rand(N) < 0.6).astype(np.uint8)
        q, _ = np.linalg.qr(rs.randn(3, 3))
        if np.linalg.det(q) < 0:
            q[:, 0] *= -1
        t = np.array([0.03 * (i % 4), -0.02, 0.8 + 0.02 * i], np.float32)
        pts = model_xyz[idx] @ q.T.astype(np.float32) + t + 0.002 * rs.randn(N, 3).astype(np.float32)
        bad = rs.rand(N) < 0.1
        pts[bad] += 0.05 * rs.randn(int(bad.sum()), 3).astype(np.float32)
        cld = np.zeros((1, 9, N), np.float32)
        cld[0, :3] = pts.T
        res = dict(mask=torch.from_numpy(mask[None]).cuda(), best_idx=torch.from_numpy(idx[None]).cuda())
        RT, valid = pose.solve_poses(res, torch.from_numpy(cld).cuda(), models[cid].model_emb.xyz)
        sel = mask.astype(bool)
        T = pose_ref.best_fit_transform(model_xyz[idx[sel]].astype(np.float64), pts[sel].astype(np.float64))
        assert bool(valid[0]) and np.abs(RT[0].cpu().numpy() - T).max() < 5e-6
        gt = np.concatenate([q, t[:, None]], axis=1).astype(np.float32)
        sub = model_xyz[:1024]
        add_g = pose.add_metric(RT, torch.from_numpy(gt[None]).cuda(), torch.from_numpy(sub).cuda()).item()
        adi_g = pose.adi_metric(RT, torch.from_numpy(gt[None]).cuda(), torch.from_numpy(sub).cuda()).item()
        Te = RT[0].cpu().numpy().astype(np.float64)
        add_w = pose_ref.add(Te[:, :3], Te[:, 3], gt[:, :3].astype(np.float64), gt[:, 3].astype(np.float64), sub.astype(np.float64))
        adi_w = pose_ref.adi(Te[:, :3], Te[:, 3], gt[:, :3].astype(np.float64), gt[:, 3].astype(np.float64), sub.astype(np.float64))
        assert abs(add_g - add_w) < 1e-6 and abs(adi_g - adi_w) < 1e-6
        assert add_w < 0.1 * ds["diameters"][cid] / 1000.0             # the ADD(-S) < 0.1 d criterion holds for a 2 mm-noise pose


def _ycbv_rank_worker(rank, world, port, out):
    """One DDP + SyncBatchNorm training step of the YCB-V configuration at its workload size, two ranks over gloo on this GPU."""
    import torch.distributed as dist
    from geometric_aware_dense_matching_amd import parallel, train_lm, train_ycb
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK="0", WORLD_SIZE=str(world))
    parallel.init_distributed("gloo")
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        args = train_ycb.build_parser().parse_args("-state=train -cls_id=21 --n-points 4096 --n-mesh 4096 --synthetic-items 8".split())
        torch.manual_seed(0)
        model = train_lm.build_model(args, 21).to(dev)
        ddp = parallel.wrap_for_training(model, local_rank=0)
        opt = torch.optim.Adam(ddp.parameters(), lr=1e-4)
        ds = train_lm.SyntheticCrops(4, 4096, 4096, seed=3)
        batch = torch.utils.data.default_collate([ds[rank * 2 + i] for i in range(2)])      # global batch 4, two crops per rank
        ddp.train()
        res, _ = train_lm.model_fn_dec(ddp, batch, dev)
        res["loss"].backward()
        grads_finite = all(torch.isfinite(p.grad).all().item() for p in ddp.parameters() if p.grad is not None)
        opt.step()
        torch.cuda.synchronize()
        chk = torch.stack([p.detach().double().sum() for p in ddp.parameters()]).cpu()
        lo, hi = chk.clone(), chk.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        ok = bool(torch.isfinite(res["loss"]).item()) and grads_finite and torch.equal(lo, hi) and bool(torch.isfinite(chk).all())
        if rank == 0:
            out.put("ok" if ok else "FAIL loss=%r finite_grads=%r in_sync=%r" % (float(res["loss"]), grads_finite, torch.equal(lo, hi)))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_config3_ycbv_training_step_at_workload_size():
    """BASELINE config 3 at ITS size: YCB-V (config/ycbv_cfg.py:100-136: 21-object table, neighbor_dis_th 0.06), object 21
    (061_foam_brick, one of the dataset's symmetric objects), N = M = 4096, batch 4, through train_ycb's own model builder.
      (1) one training step (forward, fused matching + circle loss, backward, Adam): finite loss / gradients, parameters move;
      (2) the matching loss of items 0-1 on the step's OWN descriptors vs oracle/loss_ref.py (geoMatch.py:102-157 restated on the
          CPU, radius = 0.06 * diameter(21)): value 2e-5 relative, both gradients 2e-3 relative;
      (3) the same step as two DDP + SyncBatchNorm ranks (gloo, global batch 4): finite, and parameters bit-equal across ranks."""
    import socket
    import torch.multiprocessing as mp
    from geometric_aware_dense_matching_amd import config, train_lm, train_ycb
    from oracle import loss_ref
    args = train_ycb.build_parser().parse_args("-state=train -cls_id=21 --n-points 4096 --n-mesh 4096 --synthetic-items 8".split())
    ds = config.dataset_config(args.dataset_name)
    assert args.dataset_name == "ycbv" and len(ds["objs"]) == 21 and ds["neighbor_dis_th"] == 0.06 and ds["objs"][21] in ds["sym_objs"]
    dev = torch.device("cuda")
    torch.manual_seed(0)
    model = train_lm.build_model(args, 21).to(dev).train()
    assert abs(model.positive_r - 0.06 * config.YCBV_DIAMETERS[21] / 1000.0) < 1e-9
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    data = train_lm.SyntheticCrops(4, 4096, 4096, seed=3)
    batch = torch.utils.data.default_collate([data[i] for i in range(4)])
    before = [p.detach().clone() for p in model.parameters()]
    res, cu = train_lm.model_fn_dec(model, batch, dev)
    assert res["rgbd"].shape == (4, 128, 4096) and res["mesh"].shape == (1, 128, 4096)
    res["loss"].backward()
    assert torch.isfinite(res["loss"]) and torch.isfinite(res["match_loss"]) and torch.isfinite(res["seg_loss"])
    grads = [p.grad for p in model.parameters() if p.grad is not None]
    assert len(grads) > 300 and all(torch.isfinite(g).all() for g in grads)
    opt.step()
    moved = sum(int(not torch.equal(a, p.detach())) for a, p in zip(before, model.parameters()))
    assert moved > 300
    # (2) matching loss of a sub-batch on the step's own descriptors vs the oracle
    sub = {k: cu[k][:2] for k in ("labels", "match_idx", "visible_flag", "RT")}
    rg = res["rgbd"][:2].detach().clone().requires_grad_(True)
    mg = res["mesh"].detach().clone().requires_grad_(True)
    got = model.pointwise_feature_matching(rg, mg, sub)
    got.backward()
    rc = res["rgbd"][:2].detach().cpu().clone().requires_grad_(True)
    mc = res["mesh"].detach().cpu().clone().requires_grad_(True)
    sys_idx = getattr(model.model_emb, "sys_idx", None)
    want = loss_ref.pointwise_feature_matching(rc, mc, sub["labels"].cpu().long(), sub["match_idx"].cpu().long(), sub["visible_flag"].cpu(),
                                               model.model_emb.xyz.cpu(), model.positive_r,
                                               sys_idx=sys_idx.cpu() if sys_idx is not None else None)
    want.backward()
    assert abs(got.item() - want.item()) < 2e-5 * max(1.0, abs(want.item()))
    for a, b in ((rg.grad.cpu(), rc.grad), (mg.grad.cpu(), mc.grad)):
        assert (a - b).norm().item() < 2e-3 * (b.norm().item() + 1e-20)
    del model, opt, res, cu
    torch.cuda.empty_cache()
    # (3) two ranks
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_ycbv_rank_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0
    assert out.get() == "ok"
