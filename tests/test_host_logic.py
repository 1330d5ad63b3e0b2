"""CPU: host-side mirror of the reference interface -- module tree, state_dict names/shapes, checkpoint
format, input validation.  No kernels run here."""
import json
import os

import numpy as np
import pytest
import torch

from geometric_aware_dense_matching_amd import synthetic
from geometric_aware_dense_matching_amd.config import ConfigRandLA, make_model_cfg

G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def model():
    from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
    cfg = make_model_cfg(n_mesh_node=512)
    return GeoMatch(cfg, 1, model_points=synthetic.make_model_points(1, 512))


def test_state_dict_names_and_shapes_equal_the_reference(model):
    want = json.load(open(os.path.join(G, "geomatch_state.json")))
    got = {k: list(v.shape) for k, v in model.state_dict().items() if not k.startswith("model_emb.")}
    assert set(got) == set(want)
    for k in want:
        assert got[k] == want[k], k


def test_mesh_branch_parameter_layout(model):
    sd = model.model_emb.state_dict()
    for i, cin in enumerate((9, 128, 128)):
        assert list(sd["mesh_convs.%d.weight" % i].shape) == [125, cin, 128]
        assert list(sd["mesh_convs.%d.lin.weight" % i].shape) == [128, cin]
        assert list(sd["mesh_convs.%d.bias" % i].shape) == [128]
    assert list(sd["mesh_final.weight"].shape) == [128, 9 + 3 * 128]
    for b in ("xyz", "mesh_graph_x", "mesh_graph_edge_index", "mesh_graph_edge_attr", "const_one"):
        assert b in sd
    assert list(sd["mesh_graph_edge_index"].shape) == [2, 4 * 512]


def test_reference_style_checkpoint_round_trip(model, tmp_path):
    """train_lm.py:102-146: {'epoch','model_state','optimizer_state'}, optional 'module.' prefix."""
    from geometric_aware_dense_matching_amd.checkpoint import load_checkpoint, save_checkpoint
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    path = save_checkpoint(model, opt, epoch=7, log_dir=str(tmp_path), obj_name="ape")
    assert path.endswith(os.path.join("ape", "geomatch_07.pth.tar"))
    assert os.path.exists(os.path.join(str(tmp_path), "ape", "geomatch.pth.tar"))
    ck = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ck) >= {"epoch", "model_state", "optimizer_state"}
    ck["model_state"] = {"module." + k: v for k, v in ck["model_state"].items()}      # DDP-saved form
    torch.save(ck, path)
    with torch.no_grad():
        for p in model.parameters():
            p.add_(1.0)
    ep = load_checkpoint(model, None, os.path.join(str(tmp_path), "ape", "geomatch_07"))
    assert ep == 7
    ref = torch.load(path, map_location="cpu", weights_only=False)["model_state"]
    for k, v in model.state_dict().items():
        assert torch.equal(v, ref["module." + k]), k


def test_forward_on_cpu_fails_loudly(model):
    from oracle import pyramid as opyr
    batch = synthetic.make_batch(seed=1, batch=1, n_points=1024)
    inputs = {k: torch.from_numpy(v) for k, v in batch.items()}
    pyr = opyr.build_pyramid(batch["cld_rgb_nrm"][0, :3].T.copy(), batch["dpt_xyz"][0])
    inputs.update({k: torch.from_numpy(v[None]) for k, v in pyr.items()})
    with pytest.raises(RuntimeError, match="no CPU fallback|GPU"):
        model.eval()(inputs)


def test_randla_config_defaults():
    c = ConfigRandLA()
    assert c.k_n == 16 and c.num_layers == 4 and c.in_c == 9 and list(c.d_out) == [32, 64, 128, 256]


def test_dgcnn_variant_state_dict_equals_the_reference():
    from geometric_aware_dense_matching_amd.geoMatch_DGCNN import GeoMatch as GeoMatchDGCNN
    want = json.load(open(os.path.join(G, "dgcnn_state.json")))
    m = GeoMatchDGCNN(dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, n_mesh_node=384), 1,
                      model_points=synthetic.make_model_points(1, 384))
    got = {k: list(v.shape) for k, v in m.state_dict().items()}
    assert got == want
    g = np.load(os.path.join(G, "dgcnn_eval.npz"))
    assert np.allclose(m.model_emb.mesh.numpy(), g["mesh_buffer"], atol=1e-6)     # load_mesh arithmetic (dgcnn.py:188-202)


def test_training_branch_of_conv_bn_act_blocks_on_cpu_equals_the_module_chain():
    """The training branch that hands GPU maps to the fused BatchNorm kernels must be the plain module chain everywhere else
    (CPU tensors here): same output, same gradients, same running statistics, for both block flavours."""
    from geometric_aware_dense_matching_amd import layers
    for make in (lambda: layers.pt_conv2d(6, 8, bn=True), lambda: layers.rl_conv2d(6, 8, bn=True),
                 lambda: layers.rl_conv1d(6, 8, bn=True, activation=None), lambda: layers.pt_conv1d(6, 8, bn=False)):
        torch.manual_seed(0)
        a = make().train()
        torch.manual_seed(0)
        b = make().train()
        shape = (3, 6, 5, 4) if isinstance(a.conv, torch.nn.Conv2d) else (3, 6, 20)
        x = torch.randn(*shape)
        xa, xb = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        ya = a(xa)
        yb = torch.nn.Sequential.forward(b, xb)
        ya.square().sum().backward()
        yb.square().sum().backward()
        assert torch.equal(ya, yb) and torch.equal(xa.grad, xb.grad)
        for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
            assert ka == kb and torch.equal(va, vb), ka


def test_fused_batchnorm_gate_and_host_helpers():
    from geometric_aware_dense_matching_amd import ops
    bn = torch.nn.BatchNorm2d(4).train()
    assert not ops.bn_train_supported(torch.randn(2, 4, 8, 8), bn)            # CPU tensors never take the kernel path
    assert ops._sync_group(bn) is None and ops._sync_group(torch.nn.SyncBatchNorm(4)) is None   # no process group here
    t = torch.randn(3, 5, 4, 6)
    assert torch.allclose(ops.channel_sum(t), t.sum((0, 2, 3)))
    w, x = torch.randn(7, 5), torch.randn(3, 5, 11)
    assert torch.allclose(ops.wx(w, x), torch.matmul(w, x), rtol=1e-5, atol=1e-6)
    assert ops.psp_pools_supported(32, 32) and not ops.psp_pools_supported(4, 4) and not ops.psp_pools_supported(128, 128)


def test_summarize_trace_windows(tmp_path):
    """tools/summarize_trace.py: the bench window leaves out the instrumented steps and the forward in front of the roofline loop; the
    training window is delimited by a kernel seen exactly once per step."""
    import csv
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def write(path, names):
        with open(path, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
            for i, n in enumerate(names):
                w.writerow([n, 1000 * i, 1000 * i + 500])

    step = ["void knn_kernel<1>(KnnTable)", "conv", "match_pipe_kernel"]
    bench = step * 2 + step * 4 + step * 4 + step + ["match_pipe_sim_kernel"] * 3        # warmup 2, timed 4, instrumented 4, 1 forward
    write(tmp_path / "b.csv", bench)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_trace.py"), str(tmp_path / "b.csv"), "--steps", "4"],
                         capture_output=True, text=True, check=True).stdout
    assert "4 steps" in out.splitlines()[0] and "3 kernels/step" in out.splitlines()[0]
    assert "match_pipe_sim_kernel calls=3" in out
    train = ["find_a", "find_b"] + ["fwd", "loss_once", "bwd", "adam", "adam"] * 8        # 3 warmup + 5 timed
    write(tmp_path / "t.csv", train)
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "summarize_trace.py"), str(tmp_path / "t.csv"), "--steps", "5",
                          "--periodic"], capture_output=True, text=True, check=True).stdout
    assert "5 steps" in out.splitlines()[0] and "5 kernels/step" in out.splitlines()[0]


def test_dataset_configs_match_the_reference_config_modules():
    """-dataset_name selects what the reference imports as config/<name>_cfg.py: 15 LineMOD / 21 YCB-V diameters, the neighbour
    radius factor (0.02 / 0.06), directories, object lists, batch sizes (golden: tests/golden/dataset_configs.json, dumped from
    the imported reference modules); positive_r = neighbor_dis_th * diameter / 1000 (geoMatch.py:26)."""
    import json
    from geometric_aware_dense_matching_amd import config
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "dataset_configs.json")))
    assert set(gold) == set(config.DATASET_CONFIGS) == {"lmo", "ycbv"}
    for name, g in gold.items():
        c = config.dataset_config(name)
        assert {str(k): v for k, v in c["diameters"].items()} == g["diameters"]
        assert {str(k): v for k, v in c["objs"].items()} == g["objs"]
        for k in ("neighbor_dis_th", "model_pth", "checkpoints", "model_name", "train_batch_size", "val_batch_size", "n_points", "n_mesh"):
            assert c[k] == g[k], (name, k)
        assert list(c["sym_objs"]) == g["sym_objs"]
        m = config.make_model_cfg(dataset=name)
        assert m["neighbor_dis_th"] == g["neighbor_dis_th"] and m["model_name"] == g["model_name"] and m["feat_dim"] == g["feat_dim"]
    assert len(gold["ycbv"]["diameters"]) == 21 and len(gold["lmo"]["diameters"]) == 15
    with pytest.raises(KeyError):
        config.dataset_config("tless")


def test_entry_points_select_dataset_and_variant():
    from geometric_aware_dense_matching_amd import train_lm, train_ycb
    a = train_lm.build_parser().parse_args("-state=train -cls_id=1".split())
    assert a.dataset_name == "lmo" and a.model_variant == "ffb6d" and a.batch_size is None
    b = train_ycb.build_parser().parse_args("-state=train -cls_id=21 --model-variant dgcnn".split())
    assert b.dataset_name == "ycbv" and b.model_variant == "dgcnn"
    m = train_lm.build_model(train_ycb.build_parser().parse_args("-cls_id=21 --n-mesh 64 --n-points 1024".split()), 21)
    assert abs(m.positive_r - 0.06 * 102.903 / 1000.0) < 1e-12                 # ycbv_cfg.py:25,134; geoMatch.py:26
    m = train_lm.build_model(train_lm.build_parser().parse_args("-cls_id=1 --n-mesh 64 --n-points 1024".split()), 1)
    assert abs(m.positive_r - 0.02 * 102.099 / 1000.0) < 1e-12
    with pytest.raises(KeyError):
        train_lm.build_model(train_lm.build_parser().parse_args("-cls_id=16 --n-mesh 64".split()), 16)   # not a LineMOD object
    d = train_lm.build_model(train_ycb.build_parser().parse_args("-cls_id=16 --n-mesh 64 --model-variant dgcnn".split()), 16)
    assert type(d).__module__.endswith("geoMatch_DGCNN") and d.needs_pyramid is False


def test_objects_across_gpus_shards_are_disjoint_and_cover_every_object():
    """--objects-across-gpus (train_ycb.sh:3-9 runs the 21 YCB-V objects as independent jobs): every object belongs to exactly one rank,
    for every world size the driver uses."""
    from geometric_aware_dense_matching_amd import config, train_lm
    for name, n in (("ycbv", 21), ("lmo", 8)):
        ids = list(config.dataset_config(name)["objs"])
        assert len(ids) == n
        for world in (1, 2, 4, 8):
            shards = [train_lm.object_shard(ids, r, world) for r in range(world)]
            flat = [c for s in shards for c in s]
            assert sorted(flat) == sorted(ids) and len(set(flat)) == n
            assert max(len(s) for s in shards) - min(len(s) for s in shards) <= 1
    a = train_lm.build_parser().parse_args(["--objects-across-gpus"])
    assert a.objects_across_gpus


def test_training_gemm_host_rules():
    """Host-side choices of the training GEMMs (no GPU): the K split of the pixel-contraction weight gradient divides the chunk count,
    keeps >= 4 chunks per part and about one round of workgroups; the one-pass small-channel form and the MFMA GEMM are never chosen
    for CPU tensors (the product path has no CPU fallback: those calls go to torch.bmm inside autograd only on the GPU)."""
    import torch
    from geometric_aware_dense_matching_amd import ops, settings
    for nchunk, tiles in ((48, 72), (8, 72), (768, 4), (5, 1), (1, 1), (192, 36)):
        parts = ops._wgrad_parts(nchunk, tiles)
        assert parts >= 1 and nchunk % parts == 0
        assert parts == 1 or (nchunk // parts >= 4 and parts * tiles <= 320)
    x, go = torch.zeros(2, 32, 4096), torch.zeros(2, 32, 4096)
    assert not ops.wgrad_direct_supported(x, go) and not ops.wgrad_direct_supported(x, go, bias=True)
    assert not ops.gemm_wgrad_supported(torch.zeros(2, 256, 4096), torch.zeros(2, 256, 4096))
    # every training-side switch of this round is registered (the all-switches-off parity test iterates this list)
    for name in ("USE_MFMA_GEMM_TRAIN", "USE_DIRECT_WGRAD", "USE_GATHERED_FINAL", "USE_MFMA_WGRAD", "USE_GEMM_CONV1X1_TRAIN"):
        assert isinstance(getattr(settings, name), bool)
