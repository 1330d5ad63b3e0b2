"""Template instances of the library's kernels that no other test launched (tools/kernel_coverage.py at instance level: the library's
kernel symbols against the rocprofv3 kernel list of a suite run): activation / residual / layout variants and tile shapes that are
reachable through the public arguments of an entry point but were only ever exercised in their sibling form.  Each against plain
torch in fp64 (or fp32 on the GPU for the large convolutions), tolerances as in the sibling tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from geometric_aware_dense_matching_amd import ops as o
    return o


def _act(y, act, slope):
    return y if act == 0 else torch.relu(y) if act == 1 else torch.where(y > 0, y, y * slope)


def _gen(seed):
    return torch.Generator(device="cpu").manual_seed(seed)


@pytest.mark.parametrize("act,res,aff", [(0, True, False), (0, True, True), (2, True, False), (1, True, True), (2, True, True), (0, False, False)])
def test_affine_act_every_residual_form(ops, act, res, aff):
    g = _gen(act * 4 + res * 2 + aff)
    x = torch.randn(3, 10, 6, 8, generator=g).cuda()
    r = torch.randn(3, 10, 6, 8, generator=g).cuda()
    sc, sh, rs, rb = (torch.randn(10, generator=g).cuda() for _ in range(4))
    got = ops.affine_act(x, sc, sh, act, 0.2, res=r if res else None, res_scale=rs if aff else None, res_shift=rb if aff else None, inplace=False)
    v = lambda t: t.double().view(1, 10, 1, 1)
    want = x.double() * v(sc) + v(sh)
    if res:
        want = want + (r.double() * v(rs) + v(rb) if aff else r.double())
    want = _act(want, act, 0.2)
    assert (got.double() - want).abs().max().item() < 1e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("act", [0, 1, 2])
def test_affine_act_maxk_every_activation(ops, act):
    g = _gen(act)
    x = torch.randn(2, 7, 33, 20, generator=g).cuda()
    sc, sh = torch.randn(7, generator=g).cuda(), torch.randn(7, generator=g).cuda()
    got = ops.affine_act_maxk(x, sc, sh, act, 0.2)
    want = _act(x.double() * sc.double().view(1, 7, 1, 1) + sh.double().view(1, 7, 1, 1), act, 0.2).max(dim=3)[0]
    assert (got.double() - want).abs().max().item() < 1e-6 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("pm,tpm", [(True, True), (False, True), (True, False), (False, False)])
def test_conv64_gather_add_act_mfma_every_layout_and_activation(ops, act, pm, tpm):
    g = _gen(act * 4 + pm * 2 + tpm)
    B, n, m = 2, 100, 640
    x = torch.randn(B, 64, m, generator=g).cuda()
    w = (torch.randn(64, 64, generator=g) / 8).cuda()
    t = torch.randn(B, 64, n, generator=g).cuda()
    idx = torch.randint(0, n, (B, m), generator=g).int().cuda()
    sc, sh = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
    tt = t.transpose(1, 2).contiguous() if tpm else t
    got = ops.conv64_gather_add_act_mfma(x, ops.pack_rows64(w), tt, idx, sc, sh, act, 0.2, pixel_major=pm, t_point_major=tpm)
    if pm:
        got = got.transpose(1, 2)
    want = torch.einsum("oc,bcm->bom", w.double(), x.double()) + torch.gather(t.double(), 2, idx.long().view(B, 1, m).expand(B, 64, m))
    want = _act(want * sc.double().view(1, 64, 1) + sh.double().view(1, 64, 1), act, 0.2)
    assert (got.double() - want).abs().max().item() < 3e-5 * max(1.0, want.abs().max().item())


# (B, Cin, Cout, H, W): 128-channel tiles with Cin = 64 (four k-steps) and with whole chunks; 64-channel tiles of eight and of four waves
CONV_SHAPES = [(16, 64, 512, 32, 32), (16, 128, 512, 32, 32), (16, 128, 64, 64, 64), (2, 128, 128, 32, 32), (16, 64, 64, 64, 64)]


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("res", [False, True])
@pytest.mark.parametrize("B,Cin,Cout,H,W", CONV_SHAPES)
def test_conv3x3_every_tile_shape_with_and_without_relu_and_residual(ops, act, res, B, Cin, Cout, H, W):
    g = _gen(Cin + Cout + act * 2 + res)
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    r = torch.randn(B, Cout, H, W, generator=g).cuda() if res else None
    got = ops.conv3x3_bf16x3(x, ops.conv3x3_pack_weight(w), Cout, sc, sh, act, r)
    want = torch.nn.functional.conv2d(x, w, padding=1) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)          # fp32 (MIOpen)
    if res:
        want = want + r
    want = _act(want, act, 0.0)
    assert (got - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())
    # ... and the packed output of the same launch is the pack of its fp32 output
    if (B * H * W) % 256 == 0 and Cout % 8 == 0 and (Cout == 64 or Cout % 128 == 0):
        out2, opk = ops.conv3x3_bf16x3(x, ops.conv3x3_pack_weight(w), Cout, sc, sh, act, r, out_packed=True)
        assert torch.equal(out2, got)
        written = opk.buf.clone()                                        # (the pack below may reuse the pooled buffer)
        assert torch.equal(written, ops.conv3x3_pack_act(got).buf)


# (B, Cin, Cout, n, pixel_major): Cin = 64 form; 64-channel tiles; 128-channel tiles; 144-channel tiles; pixel-major rows
GEMM_SHAPES = [(16, 64, 256, 4096, False), (2, 128, 192, 1024, False), (16, 128, 256, 4096, False), (16, 128, 576, 4096, False),
               (4, 128, 256, 1024, True)]


@pytest.mark.parametrize("act", [0, 1])
@pytest.mark.parametrize("B,Cin,Cout,n,pm", GEMM_SHAPES)
def test_gemm_every_tile_shape_with_and_without_relu(ops, act, B, Cin, Cout, n, pm):
    g = _gen(Cin + Cout + act)
    x = torch.randn(B, Cin, n, generator=g).cuda()
    w = (torch.randn(Cout, Cin, generator=g) / Cin ** 0.5).cuda()
    sc, sh = (torch.rand(Cout, generator=g) + 0.5).cuda(), torch.randn(Cout, generator=g).cuda()
    got = ops.gemm_bf16x3(x, ops.gemm_pack_weight(w), Cout, sc, sh, act, pixel_major=pm)
    want = _act(torch.matmul(w.double(), x.double()) * sc.double().view(1, -1, 1) + sh.double().view(1, -1, 1), act, 0.0)
    if pm:
        want = want.transpose(1, 2).reshape(B * n, Cout)
    assert got.shape == want.shape
    assert (got.double() - want).abs().max().item() < 3e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("act", [0, 1, 2])
def test_gather_add_affine_act_every_activation(ops, act):
    g = _gen(act)
    B, C, n, m = 2, 24, 50, 300
    x = torch.randn(B, C, m, generator=g).cuda()
    t = torch.randn(B, C, n, generator=g).cuda()
    idx = torch.randint(0, n, (B, m), generator=g).int().cuda()
    sc, sh = torch.randn(C, generator=g).cuda(), torch.randn(C, generator=g).cuda()
    want = _act(sc.double().view(1, C, 1) * (x.double() + torch.gather(t.double(), 2, idx.long().view(B, 1, m).expand(B, C, m))) + sh.double().view(1, C, 1),
                act, 0.2)
    got = ops.gather_add_affine_act(x.clone(), t, idx, sc, sh, act, 0.2)
    assert (got.double() - want).abs().max().item() < 1e-6 * max(1.0, want.abs().max().item())


def test_gather_max_with_more_than_16_neighbours(ops):
    g = _gen(1)
    B, C, n, m, K = 2, 5, 300, 500, 20
    feat = torch.randn(B, C, n, generator=g).cuda()
    idx = torch.randint(0, n, (B, m, K), generator=g).int().cuda()
    got = ops.gather_max(feat, idx)
    want = torch.gather(feat, 2, idx.long().view(B, 1, m * K).expand(B, C, m * K)).view(B, C, m, K).max(dim=3)[0]
    assert torch.equal(got, want)


def test_group_gather_backward_lds_form_with_ragged_rows(ops):
    """m K not a multiple of 4 (no 16-byte loads) and >= 4 n entries per source point: the LDS-privatised scatter, scalar loads."""
    g = _gen(2)
    B, C, n, m, K = 2, 6, 50, 77, 3
    feat = torch.randn(B, C, n, generator=g).cuda().requires_grad_(True)
    idx = torch.randint(0, n, (B, m, K), generator=g).int().cuda()
    go = torch.randn(B, C, m, K, generator=g).cuda()
    (ga,) = torch.autograd.grad(ops.group_gather(feat, idx), feat, go)
    want = torch.zeros(B, C, n, dtype=torch.float64, device="cuda")
    want.scatter_add_(2, idx.long().view(B, 1, m * K).expand(B, C, m * K), go.reshape(B, C, m * K).double())
    assert (ga.double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


def test_pointwise_many_points_few_channels_unaligned(ops):
    """> 262144 points, 16 output channels, n % 4 != 0: one wave per tile, no K split, dword operand loads."""
    g = _gen(3)
    B, n, K, Co = 2, 131074, 32, 16
    x = torch.randn(B, K, n, generator=g).cuda()
    w = (torch.randn(Co, K, generator=g) / K ** 0.5).cuda()
    got = ops.pointwise([x], w.t().contiguous())
    want = torch.einsum("ok,bkn->bon", w.double(), x.double())
    assert (got.double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


def test_pointwise_jobs_with_a_two_way_k_split(ops):
    g = _gen(4)
    B, K, Co = 2, 48, 32
    xs = [torch.randn(B, K, n, generator=g).cuda() for n in (36, 64, 9)]
    ws = [(torch.randn(K, Co, generator=g) / K ** 0.5).cuda() for _ in xs]
    outs = ops.pointwise_jobs(xs, ws)
    for x, w, o in zip(xs, ws, outs):
        assert torch.equal(o, ops.pointwise([x], w))
        want = torch.einsum("ko,bkn->bon", w.double(), x.double())
        assert (o.double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


def test_spline_layers_without_relu(ops):
    """SplineConv(..., relu=False) in its four inference forms (direct 16-byte / scalar, dense + CSR aggregation, edge-grouped) ==
    relu=True wherever the pre-activation is positive, and == the dense training-path form everywhere."""
    from geometric_aware_dense_matching_amd import _lib, splinecnn
    torch.manual_seed(6)
    M = 1024
    pos3 = torch.rand(M, 3, device="cuda")
    ei, ea = splinecnn.build_mesh_graph(pos3, k=4)
    order = torch.argsort(ei[1], stable=True)
    rowptr = torch.zeros(M + 1, dtype=torch.int32, device="cuda")
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1][order], minlength=M), 0).to(torch.int32)
    src, attr = ei[0][order].to(torch.int32).contiguous(), ea[order].contiguous()
    for cin, cout in ((9, 128), (9, 10)):
        conv = splinecnn.SplineConv(cin, cout).cuda()
        conv.bias.data.normal_(0, 0.1)
        x = torch.randn(M, cin, device="cuda")
        with torch.enable_grad():
            dense = conv(x, rowptr, src, attr, relu=False).detach()          # matmul + spline_aggregate_kernel<false>
        with torch.no_grad():
            direct = conv(x, rowptr, src, attr, relu=False)                   # spline_direct(_vec)_kernel<false>
        assert (dense - direct).abs().max().item() < 1e-5 * max(1.0, dense.abs().max().item())
        assert dense.min().item() < 0
    conv = splinecnn.SplineConv(128, 128).cuda()
    conv.bias.data.normal_(0, 0.1)
    x = torch.randn(M, 128, device="cuda")
    pairs = splinecnn.build_spline_pairs(src, attr, M)
    with torch.no_grad():
        dense = conv(x, rowptr, src, attr, relu=False)
        grouped = conv(x, rowptr, src, attr, relu=False, pairs=pairs)           # spline_pairs_aggregate_vec_kernel<false>
    assert (dense - grouped).abs().max().item() < 1e-5 * max(1.0, dense.abs().max().item()) and dense.min().item() < 0
    # the scalar pair aggregation without ReLU (C = 10), through the C entry point
    E, R, C = int(rowptr[-1]), 3000, 10
    g = _gen(7)
    Y = torch.randn(R, C, generator=g).cuda()
    pos = torch.randint(0, R, (E, 8), generator=g).int().cuda()
    basis = torch.rand(E, 8, generator=g).cuda()
    out = torch.empty(M, C, device="cuda")
    assert _lib.lib().gdm_spline_pairs_aggregate_hip(Y.data_ptr(), rowptr.data_ptr(), pos.data_ptr(), basis.data_ptr(), None, None, M, C, 0,
                                                     out.data_ptr(), None) == 0
    msg = (basis.double().unsqueeze(2) * Y.double()[pos.long()]).sum(1)
    tgt = torch.repeat_interleave(torch.arange(M, device="cuda"), (rowptr[1:] - rowptr[:-1]).long())
    want = torch.zeros(M, C, dtype=torch.float64, device="cuda").index_add_(0, tgt, msg) / (rowptr[1:] - rowptr[:-1]).double().clamp(min=1).unsqueeze(1)
    assert (out.double() - want).abs().max().item() < 1e-5 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("k", [4, 8])
def test_topk_negdist_with_few_neighbours(ops, k):
    g = _gen(k)
    B, C, n = 2, 6, 200
    x = torch.randn(B, C, n, generator=g).cuda()
    gram = torch.matmul(x.transpose(2, 1), x)
    xx = (x ** 2).sum(dim=1)
    got = ops.topk_negdist(gram, xx, k)
    pd = (-xx.view(B, 1, n) - (-2 * gram)) - xx.view(B, n, 1)                  # dgcnn.py:22-25, torch's operations in torch's order
    want = pd.topk(k=k, dim=-1)[1]
    assert torch.equal(got.long(), want)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_upconv_kernels_every_activation(ops, act):
    """The dense PSPUpsample(64, 64) kernel, the sampled-pixel last stage and the direct 9-tap gather with no activation / ReLU / PReLU."""
    g = _gen(act)
    B, H, W = 2, 16, 16
    OH, OW = 2 * H, 2 * W
    x = torch.randn(B, 64, H, W, generator=g).cuda()
    w3 = (torch.randn(64, 64, 3, 3, generator=g) * 0.05).cuda()
    sc, sh = (torch.rand(64, generator=g) + 0.5).cuda(), torch.randn(64, generator=g).cuda()
    up = torch.nn.functional.interpolate(x.double(), size=(OH, OW), mode="bilinear", align_corners=True)
    h = _act(torch.nn.functional.conv2d(up, w3.double(), padding=1) * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1), act, 0.25)
    wpk = ops.upconv_fused64_pack_weight(w3)
    got = ops.upconv_fused64(x, wpk, sc, sh, (OH, OW), act, 0.25)
    assert (got.double() - h).abs().max().item() < 3e-5 * max(1.0, h.abs().max().item())
    # sampled-pixel last stage
    N = 77
    choose = torch.randint(0, OH * OW, (B, N), generator=g).int().cuda()
    wf = (torch.randn(64, 64, generator=g) / 8).cuda()
    ref = torch.log_softmax(torch.einsum("oc,bchw->bohw", wf.double(), h), dim=1).reshape(B, 64, -1)
    ref = torch.gather(ref, 2, choose.long()[:, None, :].expand(B, 64, N))
    x_pm = x.reshape(B, 64, H * W).transpose(1, 2).contiguous()
    pts = ops.upconv_final_points(x_pm, (H, W), choose, wpk, sc, sh, act, 0.25, ops.pack_rows64(wf), None, (OH, OW))
    assert (pts.double() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item())
    # the direct gather (a scale factor the LDS tile does not fit)
    C, OH2, OW2 = 4, 20, 21
    w = (torch.randn(C, 64, 3, 3, generator=g) / 24).cuda()
    z = torch.nn.functional.conv2d(x, w.permute(2, 3, 0, 1).reshape(9 * C, 64, 1, 1).contiguous())
    up2 = torch.nn.functional.interpolate(x.double(), size=(OH2, OW2), mode="bilinear", align_corners=True)
    want = _act(torch.nn.functional.conv2d(up2, w.double(), padding=1) * sc[:C].double().view(1, -1, 1, 1) + sh[:C].double().view(1, -1, 1, 1), act, 0.25)
    out = ops.upconv3x3_gather(z, sc[:C].contiguous(), sh[:C].contiguous(), C, (OH2, OW2), act, 0.25)
    assert (out.double() - want).abs().max().item() < 1e-4 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("Cout,Cin", [(8, 24), (16, 64), (64, 16), (48, 32), (24, 24), (8, 8), (40, 40)])
def test_wgrad_direct_every_block_shape(ops, Cout, Cin):
    g = _gen(Cout * 100 + Cin)
    B, P = 3, 256
    x = torch.randn(B, Cin, P, generator=g).cuda()
    go = torch.randn(B, Cout, P, generator=g).cuda()
    gw, gb = ops.wgrad_direct(x, go, bias=True)
    want = torch.einsum("bop,bcp->oc", go.double(), x.double())
    assert (gw.double() - want).abs().max().item() < 3e-5 * max(1.0, want.abs().max().item())
    assert (gb.double() - go.double().sum((0, 2))).abs().max().item() < 1e-4 * max(1.0, go.double().sum((0, 2)).abs().max().item())


@pytest.mark.parametrize("B,H,W,with_bias", [(2, 128, 128, True), (1, 40, 24, False), (2, 13, 9, True), (1, 8, 8, True)])
def test_final_stage_at_several_sizes(ops, B, H, W, with_bias):
    """`final` = Conv2d(64, 64, 1) + LogSoftmax(dim=1) (the thread-per-pixel kernel, hardware exp) against fp64 torch, with and without
    bias, at the step's size and at ragged ones."""
    g = _gen(H * W + with_bias)
    x = torch.randn(B, 64, H, W, generator=g).cuda()
    w = (torch.randn(64, 64, 1, 1, generator=g) / 8).cuda()
    b = (torch.randn(64, generator=g) * 0.1).cuda() if with_bias else None
    want = torch.log_softmax(torch.nn.functional.conv2d(x.double(), w.double(), b.double() if with_bias else None), dim=1)
    got = ops.conv1x1_logsoftmax(x, w, b)
    assert (got.double() - want).abs().max().item() < 2e-5
