"""GPU: the training side of the path -- backward kernels inside the full network, the Trainer loop,
checkpoint round trip."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")

from geometric_aware_dense_matching_amd import settings, synthetic  # noqa: E402
from geometric_aware_dense_matching_amd.config import make_model_cfg  # noqa: E402


def test_full_network_gradients_vs_oracle_autograd():
    """d(sum(rgbd*w) + sum(seg*v))/d(params) through the HIP backward kernels (gather scatter-adds, att-pool,
    bilinear) == torch autograd through the oracle's CPU restatement.  BN uses running statistics in both
    (eval mode), so the two graphs are the same function."""
    from geometric_aware_dense_matching_amd import pyramid
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    from oracle import model_ref
    from oracle import pyramid as opyr
    M = 512
    model = GeoMatch(make_model_cfg(n_mesh_node=M), 1, model_points=synthetic.make_model_points(1, M))
    keys = json.load(open(os.path.join(G, "geomatch_state.json")))
    sd = synthetic.synthetic_state_dict({k: torch.zeros(v) for k, v in keys.items()}, seed=0)
    model.load_state_dict(sd, strict=False)
    model = model.cuda().eval()
    batch = synthetic.make_batch(seed=21, batch=1, n_points=1024)
    d = {k: torch.from_numpy(batch[k]).cuda() for k in ("rgb", "cld_rgb_nrm", "choose")}
    d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), torch.from_numpy(batch["dpt_xyz"]).cuda()))
    rs = np.random.RandomState(0)
    w = torch.from_numpy(rs.randn(1, 128, 1024).astype(np.float32))
    v = torch.from_numpy(rs.randn(1, 2, 1024).astype(np.float32))
    det, bench = torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark
    torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = True, False    # keep MIOpen off its Winograd / atomic wgrad picks
    try:
        ep = model(d)
        ((ep["rgbd"] * w.cuda()).sum() + (ep["seg"] * v.cuda()).sum()).backward()
    finally:
        torch.backends.cudnn.deterministic, torch.backends.cudnn.benchmark = det, bench

    names = ["pcd_emb.rndla_pre_stages.conv.weight", "pcd_emb.rndla_ds_stages.0.lfa.att_pooling_1.fc.weight",
             "pcd_emb.rndla_ds_stages.2.mlp1.conv.weight", "pcd_emb.ds_fuse_r2p_pre_layers.0.conv.weight",
             "pcd_emb.ds_fuse_p2r_fuse_layers.1.conv.weight", "pcd_emb.cnn_pre_stages.0.weight",
             "pcd_emb.cnn_up_stages.0.0.conv.1.weight", "pcd_emb.up_fuse_r2p_fuse_layers.2.conv.weight",
             "pcd_emb.rndla_up_stages.3.conv.weight", "feature_encoding_layer.0.conv.weight", "seg_layer.3.conv.weight"]
    sd_cpu = {k: t.clone() for k, t in sd.items()}
    for n in names:
        sd_cpu[n].requires_grad_(True)
    cpu_in = {k: torch.from_numpy(batch[k]) for k in ("rgb", "cld_rgb_nrm", "choose")}
    cpu_in.update({k: torch.from_numpy(x[None]) for k, x in
                   opyr.build_pyramid(batch["cld_rgb_nrm"][0, :3].T.copy(), batch["dpt_xyz"][0]).items()})
    out = model_ref.geomatch_forward(sd_cpu, cpu_in, torch.zeros(128, M))
    ((out["rgbd"] * w).sum() + (out["seg"] * v).sum()).backward()
    params = dict(model.named_parameters())
    for n in names:
        got, want = params[n].grad.cpu(), sd_cpu[n].grad
        # relative L2 over the tensor (MIOpen's weight-gradient algorithms differ from the CPU's in summation
        # order and, for 3x3 layers, may be Winograd: element-wise maxima are noisy, the norm is not)
        rel = (got - want).norm().item() / (want.norm().item() + 1e-20)
        assert rel < 5e-3, (n, rel)


def test_trainer_runs_saves_and_resumes(tmp_path):
    from geometric_aware_dense_matching_amd import train_lm
    from geometric_aware_dense_matching_amd.checkpoint import load_checkpoint
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    torch.manual_seed(0)
    M, N = 512, 1024
    dev = torch.device("cuda", 0)
    model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    ds = train_lm.SyntheticCrops(8, N, M, seed=3)
    loader = torch.utils.data.DataLoader(ds, batch_size=2, shuffle=False, drop_last=True, num_workers=0)
    sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=1e-6, max_lr=1e-3, cycle_momentum=False, step_size_up=4, step_size_down=4)
    bnm = train_lm.BNMomentumScheduler(model, lambda i: max(0.9 * 0.5 ** int(i * 2 / 2e5), 1e-2))
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    tr = train_lm.Trainer(model, opt, str(tmp_path), "ape", sched, bnm, dev, 0, save_every=1, log_every=2)
    n = tr.train(0, 1, loader)
    assert n == 4 and all(np.isfinite(h).all() for h in tr.history)
    changed = [k for k, v in model.named_parameters() if not torch.equal(v.detach(), before[k])]
    # every branch learns: image trunk, point branch, fusion, mesh branch, heads, loss weights
    for frag in ("cnn_pre_stages.0.weight", "rndla_ds_stages.0.lfa.att_pooling_1.fc.weight", "ds_fuse_p2r_pre_layers.0.conv.weight",
                 "model_emb.mesh_convs.0.weight", "model_emb.mesh_final.weight", "feature_encoding_layer.0.conv.weight", "awl.params"):
        assert any(frag in k for k in changed), frag
    path = os.path.join(str(tmp_path), "ape", "geomatch_00.pth.tar")
    assert os.path.exists(path) and os.path.exists(os.path.join(str(tmp_path), "ape", "geomatch.pth.tar"))
    model2 = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev)
    assert load_checkpoint(model2, None, os.path.join(str(tmp_path), "ape", "geomatch_00"), device=dev) == 0
    for (k, a), (_, b) in zip(model.state_dict().items(), model2.state_dict().items()):
        assert torch.equal(a, b), k


def test_test_entry_point_runs(tmp_path):
    from geometric_aware_dense_matching_amd import train_lm
    args = train_lm.build_parser().parse_args(("--gpus=0 -state=test -cls_id=1 --single-object --batch-size 2 --n-points 1024 --n-mesh 512 "
                                               "--synthetic-items 4 --eval-output %s" % tmp_path).split())
    res = train_lm.test(args)
    assert len(res) == 2 and res[0]["best_idx"].shape == (2, 1024) and int(res[0]["best_idx"].max()) < 512
    # the evaluator's files (evaluator.py:341,365-373,429-431): one csv line per predicted instance, R row-major, t in millimetres
    out = train_lm.test.last_outputs
    lines = open(out[0]).read().split("\n")
    assert os.path.basename(out[0]) == "ffb6d_lmo-test.csv" and lines[0] == "scene_id,im_id,obj_id,score,R,t,time" and len(lines) == 1 + 4
    f = lines[2].split(",")
    assert f[1] == "000001" and f[2:4] == ["1", "-1"] and f[6] == "-1" and int(f[0]) >= 1      # the split's own scene / image ids
    assert np.allclose(np.array(f[4].split(" "), dtype=np.float64).reshape(3, 3), res[0]["RT"][1, :, :3].double().numpy(), atol=0)
    assert np.allclose(np.array(f[5].split(" "), dtype=np.float64), 1000 * res[0]["RT"][1, :, 3].double().numpy(), atol=0)
    if len(out) > 1:                                       # the synthetic loader carries ground-truth poses: both tables were dumped
        names = [os.path.basename(p) for p in out[1:]]
        assert names == ["_lmo_test_errors.pkl", "_lmo_test_recalls.pkl", "_lmo_test_tab.txt", "ffb6d_lmo_test_errors.pkl",
                         "ffb6d_lmo_test_precisions.pkl", "ffb6d_lmo_test_tab_precisions.txt"]
        assert len(open(out[3]).read().splitlines()) == 19 and len(open(out[6]).read().splitlines()) == 18
    # a split WITHOUT scene_id / im_id: the csv is refused (loudly) instead of being filled with invented ids; the tables are still dumped
    import warnings
    saved = train_lm.SyntheticCrops.__init__.__defaults__
    train_lm.SyntheticCrops.__init__.__defaults__ = saved[:-1] + (False,)
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            train_lm.test(args)
        assert any("BOP result csv was NOT written" in str(x.message) for x in w)
        assert not any(p.endswith(".csv") for p in train_lm.test.last_outputs)
    finally:
        train_lm.SyntheticCrops.__init__.__defaults__ = saved


@pytest.mark.parametrize("entry,cls_id,extra", [("train_lm", 5, ""), ("train_ycb", 21, ""), ("train_ycb", 16, "--model-variant dgcnn")])
def test_entry_points_train_then_multi_object_test_per_dataset_and_variant(tmp_path, entry, cls_id, extra):
    """BASELINE configs 3 / 4 / 5 surfaces: `-state=train` (one epoch of two iterations, checkpoint written in the reference's
    layout under the dataset's object name) then `-state=test` with ONE MODEL PER OBJECT of the dataset (8 LM-O / 21 YCB-V,
    train_lm.py:331-340), instances dispatched by cls_id, matching + pose; the trained object's checkpoint is the one loaded."""
    import importlib
    from geometric_aware_dense_matching_amd import config, train_lm
    mod = importlib.import_module("geometric_aware_dense_matching_amd." + entry)
    common = "--n-points 1024 --n-mesh 256 --synthetic-items 4 %s" % extra
    a = mod.build_parser().parse_args(("-state=train -cls_id=%d --deterministic --batch-size 2 --epochs 1 --save-every 1 --log-every 1 --log-dir %s %s"
                                       % (cls_id, tmp_path, common)).split())
    ds = config.dataset_config(a.dataset_name)
    trainer = train_lm.train(a)
    assert len(trainer.history) == 2 and all(np.isfinite(h).all() for h in trainer.history)
    name = ds["objs"][cls_id]
    assert os.path.exists(os.path.join(str(tmp_path), name, "geomatch.pth.tar"))
    trained = {k: v.detach().clone() for k, v in trainer.model.state_dict().items()}
    t = mod.build_parser().parse_args(("-state=test -cls_id=%d -checkpoint %s --batch-size 4 %s" % (cls_id, tmp_path, common)).split())
    res = train_lm.test(t)
    n_obj = len(ds["objs"])
    assert n_obj == (8 if a.dataset_name == "lmo" else 21)
    assert len(res) == 1 and res[0]["best_idx"].shape == (4, 1024) and res[0]["RT"].shape == (4, 3, 4)
    assert res[0]["cls_id"] == sorted(ds["objs"])[:4] and int(res[0]["best_idx"].max()) < 256
    # the multi-object driver equals a per-instance pass through that instance's own model (the reference's dispatch)
    model = train_lm.build_model(t, cls_id, cache_mesh_in_eval=True).cuda()
    from geometric_aware_dense_matching_amd.checkpoint import load_checkpoint
    assert load_checkpoint(model, None, os.path.join(str(tmp_path), name, "geomatch"), device="cuda", strict=ds["load_strict"]) == 0
    for k, v in model.state_dict().items():
        assert torch.equal(v, trained[k]), k


def test_prelu1_forward_backward_equals_torch():
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(0)
    x = torch.randn(3, 5, 16, 12, device="cuda", requires_grad=True)
    a = torch.nn.Parameter(torch.tensor([0.25], device="cuda"))
    w = torch.randn_like(x)
    y = ops.prelu1(x, a)
    (y * w).sum().backward()
    gx, ga = x.grad.clone(), a.grad.clone()
    x.grad = None; a.grad = None
    yr = torch.nn.functional.prelu(x, a)
    (yr * w).sum().backward()
    assert torch.equal(y, yr)
    assert torch.equal(gx, x.grad)
    assert torch.allclose(ga, a.grad, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 24, 8, 9, 13), (1, 16, 5, 32, 32), (1, 16, 3, 40, 24)])
def test_pspupsample_lowres_training_path_matches_reference_path(B, Cin, Cout, H, W):
    """Training form of PSPUpsample: low-resolution GEMM + differentiable 9-tap gather == Upsample(x2) + Conv3x3 (outputs and all
    gradients), through train-mode BatchNorm and PReLU."""
    from geometric_aware_dense_matching_amd import cnn
    torch.manual_seed(B * 100 + Cin)
    mod = cnn.PSPUpsample(Cin, Cout).cuda().train()
    x = torch.randn(B, Cin, H, W, device="cuda", requires_grad=True)
    w = torch.randn(B, Cout, 2 * H, 2 * W, device="cuda")
    res = {}
    for flag in (False, True):
        settings.USE_LOWRES_UPCONV_TRAIN = flag
        for p in mod.parameters():
            p.grad = None
        x.grad = None
        y = mod(x)
        (y * w).sum().backward()
        res[flag] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mod.parameters()]
    settings.USE_LOWRES_UPCONV_TRAIN = True
    for a, b in zip(res[False], res[True]):
        scale = max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("B,C,H,W", [(2, 5, 32, 32), (1, 3, 7, 9), (1, 2, 64, 64)])
def test_psp_pools_backward_matches_adaptive_avg_pool(B, C, H, W):
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(H)
    x = torch.randn(B, C, H, W, device="cuda", requires_grad=True)
    ws = [torch.randn(B, C, s, s, device="cuda") for s in (1, 2, 3, 6)]
    outs = ops.psp_pools(x)
    sum((o * w).sum() for o, w in zip(outs, ws)).backward()
    got = x.grad.clone()
    x.grad = None
    refs = [torch.nn.functional.adaptive_avg_pool2d(x, s) for s in (1, 2, 3, 6)]
    for o, r in zip(outs, refs):
        assert torch.allclose(o, r, rtol=1e-5, atol=1e-6)
    sum((r * w).sum() for r, w in zip(refs, ws)).backward()
    assert torch.allclose(got, x.grad, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("s,OH,OW", [(1, 32, 32), (2, 32, 32), (3, 32, 32), (6, 32, 32), (8, 64, 64), (5, 17, 23)])
def test_upsample_bilinear_backward_small_sources(s, OH, OW):
    """The per-plane separable transpose used for the pyramid-pooling priors == autograd of F.interpolate(align_corners=True)."""
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(s)
    x = torch.randn(2, 7, s, s, device="cuda", requires_grad=True)
    w = torch.randn(2, 7, OH, OW, device="cuda")
    (ops.upsample_bilinear(x, (OH, OW)) * w).sum().backward()
    got = x.grad.clone()
    x.grad = None
    (torch.nn.functional.interpolate(x, size=(OH, OW), mode="bilinear", align_corners=True) * w).sum().backward()
    assert torch.allclose(got, x.grad, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("B,F,Cout,H,W", [(2, 16, 24, 32, 32), (1, 8, 12, 10, 14)])
def test_pspmodule_split_training_path_matches_reference_path(B, F, Cout, H, W):
    """Training form of PSPModule (split bottleneck + one-pass pools / prior sum, under autograd) == pooled priors upsampled, concatenated and
    convolved (pspnet.py:12-31): outputs and every gradient."""
    from geometric_aware_dense_matching_amd import cnn
    torch.manual_seed(B * 10 + F)
    mod = cnn.PSPModule(F, Cout).cuda().train()
    x = torch.randn(B, F, H, W, device="cuda", requires_grad=True)
    w = torch.randn(B, Cout, H, W, device="cuda")
    res = {}
    for flag in (False, True):
        settings.USE_SPLIT_PSP_TRAIN = flag
        for p in mod.parameters():
            p.grad = None
        x.grad = None
        y = mod(x)
        (y * w).sum().backward()
        res[flag] = [y.detach().clone(), x.grad.clone()] + [p.grad.clone() for p in mod.parameters()]
    settings.USE_SPLIT_PSP_TRAIN = True
    for a, b in zip(res[False], res[True]):
        scale = max(1.0, a.abs().max().item())
        assert (a - b).abs().max().item() < 2e-4 * scale


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(2, 128, 256, 32, 32), (1, 256, 128, 8, 64)])
def test_conv3x3_train_matches_autograd_of_conv2d(B, Cin, Cout, H, W):
    """Training form of the trunk's 3x3 convolutions: forward + input gradient on the split-bf16 MFMA kernel, weight gradient on MIOpen."""
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(Cin)
    x = torch.randn(B, Cin, H, W, device="cuda", requires_grad=True)
    wgt = (torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05).requires_grad_(True)
    w = torch.randn(B, Cout, H, W, device="cuda")
    y = ops.conv3x3_train(x, wgt)
    (y * w).sum().backward()
    got = [y.detach().clone(), x.grad.clone(), wgt.grad.clone()]
    x.grad = None; wgt.grad = None
    yr = torch.nn.functional.conv2d(x.double(), wgt.double(), padding=1)
    (yr * w.double()).sum().backward()
    for a, b in zip(got, [yr.detach(), x.grad, wgt.grad]):
        assert (a.double() - b.double()).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("shape,act", [((4, 16, 32, 32), 1), ((3, 8, 100), 2), ((2, 5, 12, 16), 0), ((24, 64, 64, 64), 1), ((1, 7, 8), 2)])
def test_fused_batchnorm_act_training_matches_modules(shape, act):
    """Training-mode BatchNorm + activation (gdm_bn_*): output, running statistics and all gradients == nn.BatchNorm + activation."""
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(len(shape) * 7 + act)
    C = shape[1]
    cls = torch.nn.BatchNorm2d if len(shape) == 4 else torch.nn.BatchNorm1d
    kw = dict(eps=1e-6, momentum=0.99) if act == 2 else {}
    bn_a, bn_b = cls(C, **kw).cuda().train(), cls(C, **kw).cuda().train()
    with torch.no_grad():
        bn_a.weight.copy_(torch.randn(C)); bn_a.bias.copy_(torch.randn(C))
        bn_a.running_mean.copy_(torch.randn(C)); bn_a.running_var.copy_(torch.rand(C) + 0.5)
    bn_b.load_state_dict(bn_a.state_dict())
    bn_b = bn_b.double()                                   # fp64 reference (MIOpen's fp32 sums are themselves ~1e-5 off at 10^5 elements)
    x = (torch.randn(*shape, device="cuda") * 2 + 0.7)
    w = torch.randn(*shape, device="cuda")
    xa = x.clone().requires_grad_(True)
    xb = x.double().requires_grad_(True)
    assert ops.bn_train_supported(xa, bn_a)
    ya = ops.batch_norm_act_train(xa, bn_a, act, 0.2)
    pre = bn_b(xb)
    # the activation's branch is taken from the fp32 result: where |bn(x)| is at rounding level the two precisions may pick different
    # sides, and one flipped element moves a channel's weight gradient by O(1)
    side = (ya.detach() > 0).double()
    yb = pre * side if act == 1 else pre * (side + (1 - side) * 0.2) if act == 2 else pre
    (ya * w).sum().backward()
    (yb * w.double()).sum().backward()
    keep = (pre.detach().abs() > 1e-5)
    tol = lambda t: 1e-5 * max(1.0, t.abs().max().item())
    assert ((ya.double() - yb).abs() * keep).max().item() < tol(yb)
    assert ((xa.grad.double() - xb.grad).abs() * keep).max().item() < tol(xb.grad)
    assert (bn_a.weight.grad.double() - bn_b.weight.grad).abs().max().item() < tol(bn_b.weight.grad)
    assert (bn_a.bias.grad.double() - bn_b.bias.grad).abs().max().item() < tol(bn_b.bias.grad)
    assert torch.allclose(bn_a.running_mean.double(), bn_b.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(bn_a.running_var.double(), bn_b.running_var, rtol=1e-5, atol=1e-6)
    assert int(bn_a.num_batches_tracked) == 1


def _syncbn_worker(rank, world, port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from geometric_aware_dense_matching_amd import ops
        torch.cuda.set_device(0)
        torch.manual_seed(3)
        B, C, H, W = 6, 8, 16, 24
        x = torch.randn(B, C, H, W, device="cuda") * 1.5 + 0.3
        w = torch.randn(B, C, H, W, device="cuda")
        gamma, beta = torch.randn(C, device="cuda"), torch.randn(C, device="cuda")
        # fp64 reference over the WHOLE batch
        ref = torch.nn.BatchNorm2d(C).cuda().double().train()
        with torch.no_grad():
            ref.weight.copy_(gamma); ref.bias.copy_(beta)
        xr = x.double().requires_grad_(True)
        pre = ref(xr)
        lo, hi = rank * B // world, (rank + 1) * B // world
        bn = torch.nn.SyncBatchNorm(C).cuda().train()
        with torch.no_grad():
            bn.weight.copy_(gamma); bn.bias.copy_(beta)
        xs = x[lo:hi].clone().requires_grad_(True)
        assert ops.bn_train_supported(xs, bn) and ops._sync_group(bn) is not None
        y = ops.batch_norm_act_train(xs, bn, ops.ACT_RELU)
        (y * w[lo:hi]).sum().backward()
        side = torch.zeros_like(pre)
        sides = [torch.zeros(hi - lo, C, H, W, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(sides, (y.detach() > 0).double().cpu())
        side = torch.cat(sides).cuda()
        ((pre * side) * w.double()).sum().backward()
        tol = lambda t: 1e-5 * max(1.0, t.abs().max().item())
        assert (y.double() - (pre * side)[lo:hi]).abs().max().item() < tol(pre)
        assert (xs.grad.double() - xr.grad[lo:hi]).abs().max().item() < tol(xr.grad)
        gw, gb = bn.weight.grad.double().cpu(), bn.bias.grad.double().cpu()
        dist.all_reduce(gw); dist.all_reduce(gb)                      # local sums -> whole batch (DDP would average them)
        assert (gw.cuda() - ref.weight.grad).abs().max().item() < tol(ref.weight.grad)
        assert (gb.cuda() - ref.bias.grad).abs().max().item() < tol(ref.bias.grad)
        assert torch.allclose(bn.running_mean.double(), ref.running_mean, rtol=1e-5, atol=1e-6)
        assert torch.allclose(bn.running_var.double(), ref.running_var, rtol=1e-5, atol=1e-6)
        if rank == 0:
            out.put("ok")
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_fused_syncbatchnorm_two_ranks_equals_whole_batch_batchnorm():
    """SyncBatchNorm through the fused kernels: two ranks (gloo, both on this GPU) with half the batch each == fp64 BatchNorm + ReLU over
    the whole batch -- outputs, input gradients, summed parameter gradients, running statistics."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    procs = [ctx.Process(target=_syncbn_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    assert out.get() == "ok"


def _rccl_group_of_one_worker(port, out):
    import os
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=dev)
    try:
        from geometric_aware_dense_matching_amd import ops, train_lm
        from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
        M, N, B = 512, 1024, 2
        torch.manual_seed(0)
        model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).train()
        state = {k: v.clone() for k, v in model.state_dict().items()}
        ds = train_lm.SyntheticCrops(B, N, M, seed=5)
        batch = torch.utils.data.default_collate([ds[i] for i in range(B)])

        def run(m):
            m.zero_grad(set_to_none=True)
            torch.manual_seed(1)
            o, _ = train_lm.model_fn_dec(m, batch, dev)
            o["loss"].backward()
            core = m.module if hasattr(m, "module") else m
            return float(o["loss"].detach()), {k: p.grad.detach().double().clone() for k, p in core.named_parameters() if p.grad is not None}

        l0, g0 = run(model)                                           # plain BatchNorm, no process group involved
        sync = torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)
        sync.load_state_dict(state)
        n_sync = sum(isinstance(mod, torch.nn.SyncBatchNorm) for mod in sync.modules())
        ddp = torch.nn.parallel.DistributedDataParallel(sync, device_ids=[0], output_device=0, find_unused_parameters=True)
        settings.SYNCBN_MIN_WORLD = 1                                 # the single rank goes through the collective too
        some_bn = next(mod for mod in sync.modules() if isinstance(mod, torch.nn.SyncBatchNorm))
        assert ops._sync_group(some_bn) is not None and dist.get_backend(ops._sync_group(some_bn)) == "nccl"
        l1, g1 = run(ddp)
        den = sum((v ** 2).sum().item() for v in g0.values()) ** 0.5
        d = sum(((g1[k] - g0[k]) ** 2).sum().item() for k in g0) ** 0.5 / den
        out.put((l0, l1, d, n_sync, len(g0), set(g0) == set(g1), dist.get_backend()))
    finally:
        settings.SYNCBN_MIN_WORLD = 2
        dist.destroy_process_group()


def test_training_step_under_ddp_and_syncbn_in_an_rccl_group_of_one():
    """RCCL on the card at hand: one rank joins an `nccl` process group, the model is converted to SyncBatchNorm and wrapped in
    DistributedDataParallel as parallel.wrap_for_training does for N ranks (/root/reference/train_lm.py:412,436-439), and -- with
    settings.SYNCBN_MIN_WORLD = 1 -- every fused BatchNorm sends its fp64 statistics through `dist.all_reduce` on the device in both
    directions.  A group of one changes no number: loss and gradients equal the plain single-process step (same dropout seed)."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    p = ctx.Process(target=_rccl_group_of_one_worker, args=(port, out))
    p.start()
    p.join(600)
    assert p.exitcode == 0
    l0, l1, d, n_sync, n_grads, same_keys, backend = out.get()
    print("RCCL group of one: loss %.6f vs %.6f, gradient distance %.3e, %d SyncBatchNorm layers, %d gradients" % (l0, l1, d, n_sync, n_grads))
    assert backend == "nccl" and n_sync > 80 and n_grads > 300 and same_keys
    # What "no number changes" can mean here: a randomly initialised network on a batch of 2 has channels whose batch variance is ~eps, and
    # train-mode BatchNorm amplifies ANY rounding difference between two runs by 1 / sqrt(var + eps) (see
    # test_training_step_through_fused_paths_equals_module_paths: 5e-2 .. 1e-1 between equivalent paths).  The two runs here differ in
    # the order of a few fp64 partial sums and in whatever the allocator's alignment does to kernel selection: seen 4e-5 alone in a
    # process, 9e-2 inside the whole suite.  A wrong statistic (a missing or doubled reduction, a wrong count) is an O(1) error.
    assert abs(l1 - l0) < 1e-3 * abs(l0) and d < 0.35, (l0, l1, d)


def test_training_step_through_fused_paths_equals_module_paths():
    """One whole training step (train-mode BatchNorm, dropout, fused circle loss) with the training-side kernel paths on == the same
    step with all of them switched back to the torch modules: loss, running statistics and parameter gradients.

    What the comparison can resolve: a randomly initialised network on a batch of 2 has channels whose batch variance is ~eps (RandLA's
    eps is 1e-6), and train-mode BatchNorm amplifies any rounding difference by 1/sqrt(var + eps) there.  Measured with
    tools/ab_train_paths.py (distance = relative L2 over all parameter gradients): the same path twice 2e-6; the paths that are
    exact re-associations in fp32 (low-resolution up-convolution, split PSP bottleneck) 3e-3 each; the split-bf16 convolutions (1e-5
    per output) 8e-2; the fused BatchNorm (fp64 sums against MIOpen's fp32 sums) 5e-2; everything 1.1e-1, loss 8e-5.  So the exact
    paths are held to 3e-2, and the full set to a bound that only an O(1) error -- a dropped term, a wrong scale -- would break; the
    per-operator tests above hold each kernel to 1e-4 .. 1e-5 against fp64."""
    from geometric_aware_dense_matching_amd import cnn, ops, train_lm
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    M, N, B = 512, 1024, 2
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    ds = train_lm.SyntheticCrops(B, N, M, seed=5)
    batch = torch.utils.data.default_collate([ds[i] for i in range(B)])
    flags = [(settings, "USE_LOWRES_UPCONV_TRAIN"), (settings, "USE_SPLIT_PSP_TRAIN"), (settings, "USE_MFMA_CONV_TRAIN"), (settings, "USE_FUSED_BN_TRAIN")]

    def run(on):
        for mod, name in flags:
            setattr(mod, name, name in on)
        model.load_state_dict(state)
        model.zero_grad(set_to_none=True)
        torch.manual_seed(1)                                  # same dropout masks
        out, _ = train_lm.model_fn_dec(model, batch, dev)
        out["loss"].backward()
        return (float(out["loss"].detach()),
                {k: p.grad.detach().double().clone() for k, p in model.named_parameters() if p.grad is not None},
                {k: v.double().clone() for k, v in model.state_dict().items() if k.endswith("running_var")})

    try:
        l0, g0, s0 = run(())
        l1, g1, s1 = run(("USE_LOWRES_UPCONV_TRAIN", "USE_SPLIT_PSP_TRAIN"))
        l2, g2, s2 = run(tuple(n for _, n in flags))
        # how ill-conditioned THIS step is: the module paths again on an image moved by one fp32 rounding (relative 2^-23 per
        # pixel).  Which channels have a near-zero batch variance depends on the data, so the yardstick is measured, not assumed.
        rgb0 = batch["rgb"].clone()
        sign = torch.from_numpy(np.where(np.random.RandomState(3).rand(*rgb0.shape) < 0.5, -1.0, 1.0).astype(np.float32))
        batch["rgb"] = rgb0 * (1.0 + sign * 2.0 ** -23)
        ln, gn, _ = run(())
        batch["rgb"] = rgb0
    finally:
        for mod, name in flags:
            setattr(mod, name, True)
    den = sum((v ** 2).sum().item() for v in g0.values()) ** 0.5
    dist = lambda g: sum(((g[k] - g0[k]) ** 2).sum().item() for k in g0) ** 0.5 / den
    noise = dist(gn)
    print("training A/B: loss %.6f / %.6f / %.6f, gradient distance exact paths %.3e, all paths %.3e, one-rounding input noise %.3e"
          % (l0, l1, l2, dist(g1), dist(g2), noise))
    assert np.isfinite(l0) and len(g0) > 300 and set(g0) == set(g1) == set(g2)
    assert abs(l1 - l0) < 1e-4 * abs(l0) and abs(l2 - l0) < 1e-3 * abs(l0), (l0, l1, l2)
    for k in s0:
        assert torch.allclose(s0[k], s1[k], rtol=5e-2, atol=1e-3) and torch.allclose(s0[k], s2[k], rtol=5e-2, atol=1e-3), k
    # fp32 re-associations must stay within what ONE input rounding already does to this step (x10 for the number of re-associated
    # sums), the split-bf16 / fp64-sum paths within a bound that only an O(1) error would break
    assert dist(g1) < max(3e-2, 10.0 * noise), (dist(g1), noise)
    assert dist(g2) < 0.35, dist(g2)


@pytest.mark.parametrize("shape,cout,bias", [((3, 16, 50), 24, True), ((2, 32, 40, 16), 32, False), ((2, 64, 24, 24), 64, True)])
def test_conv1x1_train_as_batched_gemms_equals_torch_convolution(shape, cout, bias):
    """ops.conv1x1_train (forward W.x, input gradient W^T.go, weight gradient sum_b go.x^T as batched GEMMs) == the nn.Conv1d / nn.Conv2d
    module under autograd (fp64), followed by an in-place activation as the layer modules apply one."""
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(cout)
    cls = torch.nn.Conv1d if len(shape) == 3 else torch.nn.Conv2d
    conv = cls(shape[1], cout, 1, bias=bias).cuda()
    x = torch.randn(*shape, device="cuda", requires_grad=True)
    w = torch.randn(shape[0], cout, *shape[2:], device="cuda")
    assert ops.conv1x1_train_supported(conv, x)
    y = torch.nn.functional.leaky_relu_(ops.conv1x1_train(conv, x), 0.2)
    (y * w).sum().backward()
    got = [y.detach().double(), x.grad.double(), conv.weight.grad.double()] + ([conv.bias.grad.double()] if bias else [])
    c64 = cls(shape[1], cout, 1, bias=bias).cuda().double()
    c64.load_state_dict({k: v.double() for k, v in conv.state_dict().items()})
    x64 = x.detach().double().requires_grad_(True)
    y64 = torch.nn.functional.leaky_relu(c64(x64), 0.2)
    (y64 * w.double()).sum().backward()
    want = [y64.detach(), x64.grad, c64.weight.grad] + ([c64.bias.grad] if bias else [])
    for a, b in zip(got, want):
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("B,Cin,Cout,H,W,parts", [(2, 256, 256, 32, 32, None), (3, 256, 64, 32, 32, 4), (1, 512, 136, 8, 64, 1), (2, 256, 256, 64, 64, None)])
def test_conv3x3_weight_gradient_on_the_mfma_gemm_equals_autograd(B, Cin, Cout, H, W, parts):
    """ops.conv3x3_wgrad (pixels as the contraction axis of the split-bf16 MFMA GEMM, nine tap-shifted operand copies, K split in parts)
    == d/dW of conv2d(x, W, padding=1) under fp64 autograd: relative L2 3e-5, max 1e-4 of the gradient's scale."""
    from geometric_aware_dense_matching_amd import ops
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout + H)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    go = torch.randn(B, Cout, H, W, generator=g).cuda()
    assert ops.conv3x3_wgrad_supported(x, go)
    got = ops.conv3x3_wgrad(x, go, parts=parts)
    w = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, device="cuda", requires_grad=True)
    (torch.nn.functional.conv2d(x.double(), w, padding=1) * go.double()).sum().backward()
    want = w.grad
    assert got.shape == want.shape
    assert (got.double() - want).norm().item() < 3e-5 * want.norm().item()
    assert (got.double() - want).abs().max().item() < 1e-4 * want.abs().max().item()


@pytest.mark.parametrize("B,Cin,Cout,P", [(2, 256, 256, 4096), (3, 1024, 2304, 256), (1, 512, 72, 16384), (24, 1024, 2304, 256)])
def test_pixel_contraction_weight_gradient_of_a_1x1_product_equals_fp64(B, Cin, Cout, P):
    """ops.gemm_wgrad (dW = sum_b go[b] . x[b]^T with the pixels as the contraction axis of the split-bf16 MFMA GEMM) == the fp64 product:
    relative L2 3e-5, max 1e-4 of the gradient's scale."""
    from geometric_aware_dense_matching_amd import ops
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout + P)
    x = torch.randn(B, Cin, P, generator=g).cuda()
    go = torch.randn(B, Cout, P, generator=g).cuda()
    got = ops.gemm_wgrad(x, go)
    want = torch.bmm(go.double(), x.double().transpose(1, 2)).sum(0)
    assert got.shape == want.shape
    assert (got.double() - want).norm().item() < 3e-5 * want.norm().item()
    assert (got.double() - want).abs().max().item() < 1e-4 * want.abs().max().item()


@pytest.mark.parametrize("B,Cin,Cout,n,conv", [(4, 1024, 2304, 256, False), (2, 512, 256, 4096, False), (2, 256, 512, 4096, True), (2, 64, 32, 1000, False)])
def test_wx_training_products_on_the_mfma_gemm_equal_fp64_autograd(B, Cin, Cout, n, conv):
    """ops.wx / ops.conv1x1_train under autograd (forward, input gradient, weight gradient each on the split-bf16 MFMA GEMM where large,
    on a batched fp32 GEMM otherwise) == fp64 autograd of the same product, followed by an in-place activation (a smooth one: at a kink a 1e-5
    difference of the product picks the other slope for that element): 1e-4 of each scale."""
    from geometric_aware_dense_matching_amd import ops
    torch.manual_seed(Cin + n)
    x = torch.randn(B, Cin, n, device="cuda", requires_grad=True)
    wgt = torch.randn(B, Cout, n, device="cuda")
    if conv:
        m = torch.nn.Conv1d(Cin, Cout, 1, bias=True).cuda()
        y = torch.tanh_(ops.conv1x1_train(m, x))
        w = m.weight
    else:
        w = (torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5).requires_grad_(True)
        y = torch.tanh(ops.wx(w, x))
    (y * wgt).sum().backward()
    got = [y.detach().double(), x.grad.double(), w.grad.double().reshape(Cout, Cin)]
    x64 = x.detach().double().requires_grad_(True)
    w64 = w.detach().double().reshape(Cout, Cin).requires_grad_(True)
    y64 = torch.matmul(w64, x64)
    if conv:
        y64 = y64 + m.bias.detach().double().view(1, -1, 1)
    y64 = torch.tanh(y64)
    (y64 * wgt.double()).sum().backward()
    for a, b in zip(got, [y64.detach(), x64.grad, w64.grad]):
        assert (a - b).abs().max().item() < 1e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("B,Cin,Cout,P,sliced", [(3, 32, 32, 65536, False), (2, 64, 128, 16384, True), (24, 128, 128, 4096, False), (2, 40, 24, 8192, False),
                                                 (1, 8, 3, 16384, True), (5, 64, 64, 4128, False)])
def test_direct_weight_and_bias_gradient_of_small_channel_layers_equals_fp64(B, Cin, Cout, P, sliced):
    """ops.wgrad_direct (one pass over the fp32 rows, fragments split to bf16 hi / lo in registers, K split over workgroups) ==
    the fp64 products; `sliced`: the operands are channel slices of wider tensors (batch stride != C * P), as the gradient of a
    concatenation hands them over.  Relative L2 3e-5, max 1e-4 of the scale."""
    from geometric_aware_dense_matching_amd import ops
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout + P)
    xw = torch.randn(B, Cin + (16 if sliced else 0), P, generator=g).cuda()
    gw_ = torch.randn(B, Cout + (8 if sliced else 0), P, generator=g).cuda()
    x, go = xw[:, 4:4 + Cin] if sliced else xw, gw_[:, 8:8 + Cout] if sliced else gw_
    assert ops.wgrad_direct_supported(x, go, bias=True)
    got_w, got_b = ops.wgrad_direct(x, go, bias=True)
    want_w = torch.bmm(go.double(), x.double().transpose(1, 2)).sum(0)
    want_b = go.double().sum((0, 2))
    assert got_w.shape == want_w.shape and got_b.shape == want_b.shape
    assert (got_w.double() - want_w).norm().item() < 3e-5 * want_w.norm().item()
    assert (got_w.double() - want_w).abs().max().item() < 1e-4 * want_w.abs().max().item()
    assert (got_b.double() - want_b).abs().max().item() < 1e-4 * max(1.0, want_b.abs().max().item())
