"""GPU: the training iteration as one hipGraph launch (geometric_aware_dense_matching_amd/train_graph.py) -- the static-shape form of
the matching loss it needs, and the captured iteration against the eager loop of /root/reference/train_lm.py:224-296."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from geometric_aware_dense_matching_amd import settings, synthetic  # noqa: E402
from geometric_aware_dense_matching_amd.config import make_model_cfg  # noqa: E402


def _model(M, N, seed=0):
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    torch.manual_seed(seed)
    m = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).cuda().train()
    m.model_emb.dropout = 0.0                                    # eager and captured RNG streams differ; everything else is the same kernels
    return m


@pytest.mark.parametrize("symmetric", [False, True])
def test_matching_loss_over_all_rows_with_zero_weights_equals_compacted_rows(symmetric):
    """settings.STATIC_MATCH_ROWS (what a capture uses): value and both gradients equal the compacted form's, including an item with
    fewer than 3 selected points (dropped, geoMatch.py:126-127), rows without correspondence (match == M)."""
    B, N, M = 3, 512, 512
    model = _model(M, N)
    if symmetric:
        model.model_emb.set_symmetry(torch.from_numpy(np.random.RandomState(5).permutation(M)).cuda())
    rs = np.random.RandomState(1)
    labels = (rs.rand(B, N) < 0.5).astype(np.int32)
    labels[1] = 0
    labels[1, :2] = 1                                            # 2 selected points: the item is skipped
    match = rs.randint(0, M + 1, size=(B, N)).astype(np.int32)
    x = dict(labels=torch.from_numpy(labels).cuda(), match_idx=torch.from_numpy(match).cuda(),
             visible_flag=torch.from_numpy((rs.rand(B, M) < 0.6).astype(np.float32)).cuda())
    f0 = torch.from_numpy(rs.randn(B, 128, N).astype(np.float32)).cuda()
    m0 = torch.from_numpy(rs.randn(1, 128, M).astype(np.float32)).cuda()
    res = []
    try:
        for static in (False, True):
            settings.STATIC_MATCH_ROWS = static
            f, m = f0.clone().requires_grad_(True), m0.clone().requires_grad_(True)
            loss = model.pointwise_feature_matching(f, m, x)
            loss.backward()
            res.append((loss.item(), f.grad.clone(), m.grad.clone()))
    finally:
        settings.STATIC_MATCH_ROWS = False
    (l0, gf0, gm0), (l1, gf1, gm1) = res
    assert np.isfinite(l0) and abs(l0 - l1) < 1e-6 * abs(l0)
    assert gf0[1].abs().max().item() == 0 and gf1[1].abs().max().item() == 0          # the skipped item gets no gradient
    assert (gf1 - gf0).abs().max().item() < 1e-6 * gf0.abs().max().item() + 1e-9
    assert (gm1 - gm0).abs().max().item() < 2e-5 * gm0.abs().max().item()             # atomics: summation order differs


def test_no_selected_rows_gives_zero_loss_in_both_forms():
    B, N, M = 2, 256, 256
    model = _model(M, N)
    x = dict(labels=torch.zeros(B, N, dtype=torch.int32).cuda(), match_idx=torch.zeros(B, N, dtype=torch.int32).cuda(),
             visible_flag=torch.ones(B, M).cuda())
    f = torch.randn(B, 128, N).cuda().requires_grad_(True)
    m = torch.randn(1, 128, M).cuda().requires_grad_(True)
    try:
        for static in (False, True):
            settings.STATIC_MATCH_ROWS = static
            loss = model.pointwise_feature_matching(f, m, x)
            assert loss.item() == 0.0
    finally:
        settings.STATIC_MATCH_ROWS = False


def test_graphed_training_iterations_equal_the_eager_loop():
    """Six optimiser steps on six different batches: GraphedTrainStep (3 eager warm-up iterations on its side stream, one capture,
    3 replays) against the plain loop model_fn_dec -> backward -> Adam.step -> zero_grad.  Same kernels in the same order on both
    sides; what differs is the summation order of the atomics, which train-mode BatchNorm on a small batch amplifies (see
    test_training_step_through_fused_paths_equals_module_paths), so the loss trajectory is held to 10x what the eager loop differs from ITSELF run twice, and the
    parameters to the size of the Adam steps taken.  A cyclic learning rate is stepped between iterations on both sides (the capture reads it from the device)."""
    from geometric_aware_dense_matching_amd import train_lm
    from geometric_aware_dense_matching_amd.train_graph import GraphedTrainStep
    M, N, B, steps = 512, 1024, 4, 6
    dev = torch.device("cuda", 0)
    ds = train_lm.SyntheticCrops(B * steps, N, M, seed=3)
    batches = [torch.utils.data.default_collate([ds[s * B + i] for i in range(B)]) for s in range(steps)]

    def run(graphed):
        model = _model(M, N, seed=7)
        opt = torch.optim.Adam(model.parameters(), lr=1e-4)
        stepper = GraphedTrainStep(model, opt, dev) if graphed else None
        sched = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=1e-7, max_lr=1e-5, cycle_momentum=False, step_size_up=4, step_size_down=4)
        losses, lrs = [], []
        for b in batches:
            lrs.append(float(opt.param_groups[0]["lr"]))
            if graphed:
                out = stepper.step(b)
            else:
                out, _ = train_lm.model_fn_dec(model, b, dev)
                out["loss"].backward()
                opt.step()
                opt.zero_grad()
            losses.append([float(torch.as_tensor(out[k]).detach()) for k in ("loss", "seg_loss", "match_loss")])
            sched.step()
        return np.array(losses), lrs, {k: v.detach().double().clone() for k, v in model.state_dict().items()}, stepper

    le, lre, pe, _ = run(False)
    le2, _, pe2, _ = run(False)                                  # the yardstick: the eager loop against itself (atomics' summation order)
    lg, lrg, pg, stepper = run(True)
    assert stepper.captures == 1 and stepper.calls == steps
    assert np.allclose(lre, lrg, rtol=1e-6) and len(set(lre)) >= steps - 1  # the schedule moved, and the capture followed it
    noise = np.abs(le2 - le).max(axis=1)
    diff = np.abs(lg - le).max(axis=1)
    print("eager losses  ", le[:, 0], "\ngraphed losses", lg[:, 0], "\neager-vs-eager", noise, "\ngraph-vs-eager", diff)
    assert np.isfinite(lg).all()
    assert (diff <= np.maximum(10.0 * noise, 2e-4 * np.abs(le).max(axis=1))).all(), (diff, noise)
    names = [k for k, _ in _model(M, N, seed=7).named_parameters()]
    for k in pe:
        if k.endswith("num_batches_tracked"):
            assert torch.equal(pe[k], pg[k]) and int(pe[k]) == steps, k
        elif k.endswith("running_mean") or k.endswith("running_var"):
            assert torch.allclose(pe[k], pg[k], rtol=5e-2, atol=1e-3), k
    moved_e = max((pe2[k] - pe[k]).abs().max().item() for k in names)
    moved_g = max((pg[k] - pe[k]).abs().max().item() for k in names)
    took = max((pe[k] - _model(M, N, seed=7).state_dict()[k].double()).abs().max().item() for k in names[:20])
    print("parameters: eager-vs-eager %.3e, graph-vs-eager %.3e, moved by training %.3e" % (moved_e, moved_g, took))
    assert took > 1e-6                                            # the optimiser did step
    assert moved_g <= 2 * sum(lre) + 1e-9                         # never further apart than the steps taken (|Adam step| ~ lr, either sign)


def test_bn_momentum_change_recaptures():
    from geometric_aware_dense_matching_amd import train_lm
    from geometric_aware_dense_matching_amd.train_graph import GraphedTrainStep
    M, N, B = 256, 1024, 2
    dev = torch.device("cuda", 0)
    model = _model(M, N)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    stepper = GraphedTrainStep(model, opt, dev, warmup=2)
    ds = train_lm.SyntheticCrops(B, N, M, seed=9)
    batch = torch.utils.data.default_collate([ds[i] for i in range(B)])
    for _ in range(4):
        out = stepper.step(batch)
    assert stepper.captures == 1
    bnm = train_lm.BNMomentumScheduler(model, lambda i: 0.05)
    bnm.step()
    out = stepper.step(batch)
    assert stepper.captures == 2 and np.isfinite(float(out["loss"]))
    with pytest.raises(ValueError):
        stepper.step({k: v[:1] for k, v in batch.items()})       # a short last batch is rejected, not silently mis-run
