"""Random small / odd shapes through the main operators against torch references (development aid: looks for latent faults)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_aware_dense_matching_amd import ops, randla, settings
rs = np.random.RandomState(int(os.environ.get("SEED", "0")))
torch.manual_seed(int(os.environ.get("SEED", "0")))
bad = 0
def chk(name, ok, info):
    global bad
    if not ok:
        bad += 1
        print("MISMATCH", name, info, flush=True)
# match
for it in range(40):
    B, N, M = rs.randint(1, 4), rs.randint(1, 700), rs.randint(1, 900)
    if it % 4 == 0: N, M = 256 * rs.randint(1, 4), 256 * rs.randint(1, 4)
    s = torch.randn(B, 128, N, device="cuda"); m = torch.randn(128, M, device="cuda")
    gi, gv, gs = ops.match(s, m, precision=0, return_sim=True)
    sn = torch.nn.functional.normalize(s, dim=1); mn = torch.nn.functional.normalize(m, dim=0)
    ref = torch.einsum("bdn,dm->bnm", sn.double(), mn.double())
    chk("match", (gs.double() - ref).abs().max().item() < 1e-4 and (gv.double() - ref.max(dim=2)[0]).abs().max().item() < 1e-4, (B, N, M))
# knn
for it in range(40):
    B, S, Q, K = rs.randint(1, 4), rs.randint(1, 3000), rs.randint(1, 600), int(rs.choice([1, 3, 8, 16, 20, 32]))
    sup = torch.rand(B, S, 3, device="cuda"); q = torch.rand(B, Q, 3, device="cuda")
    idx, d2 = ops.knn_batch(sup, q, K, return_d2=True)
    dm = ((q[:, :, None, :] - sup[:, None, :, :]) ** 2).sum(-1)
    kk = min(K, S)
    rv = torch.topk(dm, kk, dim=2, largest=False)[0]
    got = torch.gather(dm, 2, idx[:, :, :kk].long())
    chk("knn", torch.allclose(got, rv, rtol=1e-5, atol=1e-7), (B, S, Q, K))
# gemm
for it in range(30):
    B, Cin, Cout, n = rs.randint(1, 3), int(rs.choice([64, 128, 256, 384])), rs.randint(1, 700), 32 * rs.randint(1, 40)
    if not ops.gemm_supported(Cin, Cout, n): continue
    x = torch.randn(B, Cin, n, device="cuda"); w = torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5
    got = ops.gemm_bf16x3(x, ops.gemm_pack_weight(w), Cout)
    ref = torch.matmul(w.double(), x.double())
    chk("gemm", (got.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (B, Cin, Cout, n))
# conv3x3
for it in range(12):
    B, Cin, Cout, H, W = rs.randint(1, 3), int(rs.choice([128, 256])), int(rs.choice([128, 256])), rs.randint(1, 40), 32 * rs.randint(1, 3)
    x = torch.randn(B, Cin, H, W, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    got = ops.conv3x3_bf16x3(x, ops.conv3x3_pack_weight(w), Cout)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    chk("conv3x3", (got.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (B, Cin, Cout, H, W))
# LFA blocks
for it in range(16):
    d_out = int(rs.choice([32, 64, 128, 256])); n = rs.randint(1, 300); B = rs.randint(1, 3)
    blk = randla.BuildingBlock(d_out).cuda().eval()
    xyz = torch.randn(B, n, 3, device="cuda"); feat = torch.randn(B, d_out // 2, n, 1, device="cuda")
    idx = torch.randint(0, n, (B, n, 16), device="cuda", dtype=torch.int32)
    with torch.no_grad():
        settings.USE_FUSED_LFA = False; ref = blk(xyz, feat, idx)
        settings.USE_FUSED_LFA = True; got = blk(xyz, feat, idx)
    chk("lfa", (got - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (d_out, n, B))
# gather_max: every dispatch form (LDS-staged rows, thread per (channel, point), point per thread)
for it in range(40):
    B, C, n, m, K = rs.randint(1, 4), rs.randint(1, 300), int(rs.choice([7, 100, 1024, 1500, 4096, 16384, 17000])), rs.randint(1, 700), int(rs.choice([1, 5, 16, 20]))
    feat = torch.randn(B, C, n, device="cuda"); idx = torch.randint(0, n, (B, m, K), device="cuda", dtype=torch.int32)
    got = ops.gather_max(feat, idx)
    ref = torch.gather(feat[:, :, None, :].expand(B, C, m, n), 3, idx.long()[:, None].expand(B, C, m, K)).max(dim=3)[0] if C * m * n < 3e8 else None
    if ref is not None:
        chk("gather_max", torch.equal(got, ref), (B, C, n, m, K))
# strided / plain 3x3 and 1x1 on one packed operand
for it in range(12):
    B, Cin, Cout, Ho, Wo = rs.randint(1, 3), int(rs.choice([64, 128, 256])), int(rs.choice([64, 128, 200, 256])), rs.randint(1, 20), 32 * rs.randint(1, 3)
    if Cout % 8: continue
    x = torch.randn(B, Cin, 2 * Ho, 2 * Wo, device="cuda"); w = torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5
    xp = ops.conv3x3_pack_act(x)
    got = ops.conv3x3_bf16x3(xp, ops.conv3x3_pack_weight(w), Cout, stride=2)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=1)
    chk("conv3x3 s2", (got.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (B, Cin, Cout, Ho, Wo))
    if ops.gemm_supported(Cin, Cout, 64):
        w1 = torch.randn(Cout, Cin, device="cuda") / Cin ** 0.5
        got1 = ops.conv1x1_packed2d(xp, ops.gemm_pack_weight(w1), Cout, stride=2)
        ref1 = torch.nn.functional.conv2d(x.double(), w1.double()[:, :, None, None], stride=2)
        chk("conv1x1 s2", (got1.double() - ref1).abs().max().item() < 3e-5 * max(1.0, ref1.abs().max().item()), (B, Cin, Cout, Ho, Wo))
# the sampled-pixel final stage and the per-point heads at ragged sizes
for it in range(12):
    B, H, W, N = rs.randint(1, 4), rs.randint(2, 40), rs.randint(2, 40), rs.randint(1, 300)
    x = torch.randn(B, 64, H, W, device="cuda"); w3 = torch.randn(64, 64, 3, 3, device="cuda") * 0.05; wf = torch.randn(64, 64, device="cuda") / 8
    sc = torch.rand(64, device="cuda") + 0.5; sh = torch.randn(64, device="cuda"); bf = torch.randn(64, device="cuda")
    ch = torch.randint(0, 4 * H * W, (B, N), device="cuda", dtype=torch.int32)
    up = torch.nn.functional.interpolate(x.double(), size=(2 * H, 2 * W), mode="bilinear", align_corners=True)
    h = torch.nn.functional.conv2d(up, w3.double(), padding=1) * sc.double()[None, :, None, None] + sh.double()[None, :, None, None]
    h = torch.where(h > 0, h, 0.25 * h)
    ref = torch.log_softmax(torch.einsum("oc,bchw->bohw", wf.double(), h) + bf.double()[None, :, None, None], dim=1).reshape(B, 64, -1)
    ref = torch.gather(ref, 2, ch.long()[:, None, :].expand(B, 64, N))
    got = ops.upconv_final_points(x.reshape(B, 64, H * W).transpose(1, 2).contiguous(), (H, W), ch, ops.upconv_fused64_pack_weight(w3), sc, sh, 2, 0.25,
                                  ops.pack_rows64(wf), bf, (2 * H, 2 * W))
    chk("final_points", (got.double() - ref).abs().max().item() < 5e-5 * max(1.0, ref.abs().max().item()), (B, H, W, N))
for it in range(8):
    B, N, Ca = rs.randint(1, 4), rs.randint(1, 400), int(rs.choice([8, 64, 120, 128]))
    x0 = torch.randn(B, 128, N, device="cuda")
    Ws = [torch.randn(128, 128, device="cuda") / 11 for _ in range(8)]
    scs = [torch.rand(128, device="cuda") + 0.5 for _ in range(8)]; shs = [torch.randn(128, device="cuda") * 0.3 for _ in range(8)]
    acts = [1, 1, 1, 0, 1, 1, 1, 1]
    wl = torch.randn(2, 128, device="cuda") / 11; bl = torch.randn(2, device="cuda")
    xx = x0.double(); feat = None
    for l in range(8):
        y = torch.einsum("oc,bcn->bon", Ws[l].double(), xx) * scs[l].double()[None, :, None] + shs[l].double()[None, :, None]
        if l == 3: feat = y
        if acts[l]: y = y.clamp(min=0)
        if l == 4: y = x0.double() + y
        xx = y
    seg = torch.einsum("oc,bcn->bon", wl.double(), xx) + bl.double()[None, :, None]
    layers = [(ops.gemm_pack_weight(Ws[l]), scs[l], shs[l], acts[l]) for l in range(8)]
    gf, gs = ops.point_heads(x0[:, :Ca].contiguous(), x0[:, Ca:].contiguous() if Ca < 128 else None, layers, (ops.gemm_pack_weight(wl), bl, 2), 3, 4)
    chk("heads", (gf.double() - feat).abs().max().item() < 5e-5 * max(1.0, feat.abs().max().item())
        and (gs.double() - seg).abs().max().item() < 1e-4 * max(1.0, seg.abs().max().item()), (B, N, Ca))
# round 3: the kernels added in the second half of the round, at ragged sizes
for it in range(16):                                             # LFA stages on the fp32 MFMA vs the separate-kernel chain
    d_out, n, B = int(rs.choice([32, 64, 128, 256])), rs.randint(1, 300), rs.randint(1, 4)
    torch.manual_seed(it)
    blk = randla.BuildingBlock(d_out).cuda().eval()
    for mod in blk.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.5); mod.running_var.uniform_(0.5, 2.0); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.3)
    xyz = torch.randn(B, n, 3, device="cuda"); feat = torch.randn(B, d_out // 2, n, 1, device="cuda")
    idx = torch.randint(0, n, (B, n, 16), device="cuda", dtype=torch.int32)
    with torch.no_grad():
        settings.USE_FUSED_LFA = False; ref = blk(xyz, feat, idx)
        settings.USE_FUSED_LFA = True; got = blk(xyz, feat, idx)
    chk("lfa", (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), (d_out, n, B))
for it in range(24):                                             # one-pass weight + bias gradient, sliced operands
    B, Cin, Cout, P = rs.randint(1, 6), rs.randint(1, 129), rs.randint(1, 129), 32 * rs.randint(16, 600)
    if B * P < 16384: P = 32 * (16384 // (32 * B) + 1)
    xw = torch.randn(B, Cin + 4, P, device="cuda"); gw = torch.randn(B, Cout + 8, P, device="cuda")
    x, go = xw[:, 4:], gw[:, :Cout]
    if not ops.wgrad_direct_supported(x, go, bias=True): continue
    w, b_ = ops.wgrad_direct(x, go, bias=True)
    rw_ = torch.bmm(go.double(), x.double().transpose(1, 2)).sum(0); rb = go.double().sum((0, 2))
    chk("wgrad_direct", (w.double() - rw_).norm().item() < 3e-5 * rw_.norm().item() and (b_.double() - rb).abs().max().item() < 1e-4 * max(1.0, rb.abs().max().item()), (B, Cin, Cout, P))
for it in range(10):                                             # producers that write the packed operand themselves
    B, H = rs.randint(1, 5), int(rs.choice([8, 16, 32]))
    C, W = int(rs.choice([64, 128, 256])), 32
    if (B * H * W) % 256: continue
    g = torch.randn(B, C, H, W, device="cuda"); ys = [torch.randn(B, C, s_, s_, device="cuda") for s_ in (1, 2, 3, 6)]; bias = torch.randn(C, device="cuda")
    a = ops.psp_combine(g.clone(), ys, bias); b2 = ops.psp_combine(g.clone(), ys, bias, packed=True)
    ok = torch.equal(a, b2) and torch.equal(ops.conv3x3_pack_act(b2.clone()).buf, b2._gdm_packed.buf.clone())
    z = torch.randn(B, 9 * C, H // 2, W // 2, device="cuda"); sc = torch.rand(C, device="cuda") + 0.5; sh = torch.randn(C, device="cuda")
    a = ops.upconv3x3_gather(z, sc, sh, C, (H, W), 2, 0.25); b2 = ops.upconv3x3_gather(z, sc, sh, C, (H, W), 2, 0.25, packed=True)
    ok = ok and torch.equal(a, b2) and torch.equal(ops.conv3x3_pack_act(b2.clone()).buf, b2._gdm_packed.buf.clone())
    chk("packed_producers", ok, (B, C, H, W))
for it in range(8):                                              # 3x3 convolution, 128 -> 128: half tiles at small batch
    B, H = rs.randint(1, 6), int(rs.choice([8, 32]))
    if (B * H * 32) % 128: continue
    x = torch.randn(B, 128, H, 32, device="cuda"); w = torch.randn(128, 128, 3, 3, device="cuda") / 34
    got = ops.conv3x3_bf16x3(x, ops.conv3x3_pack_weight(w), 128)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
    chk("conv_half_tiles", (got.double() - ref).abs().max().item() < 3e-5 * max(1.0, ref.abs().max().item()), (B, H))
# round 4: per-point layers in every dispatch form (FMA for K < 32, K-split MFMA, many-point channel-wave form), both weight layouts
for it in range(30):
    B, K, Co = rs.randint(1, 4), int(rs.choice([3, 9, 16, 32, 48, 64, 130, 256])), int(rs.choice([1, 8, 16, 24, 64, 100, 128]))
    n = int(rs.choice([1, 7, 64, 1000, 4096, 30000, 70001]))
    x = torch.randn(B, K, n, device="cuda"); w = torch.randn(Co, K, device="cuda") / K ** 0.5
    sc = torch.rand(Co, device="cuda") + 0.5; sh = torch.randn(Co, device="cuda")
    ref = torch.einsum("ok,bkn->bon", w.double(), x.double()) * sc.double()[None, :, None] + sh.double()[None, :, None]
    ref = torch.where(ref > 0, ref, ref * 0.2)
    a = ops.pointwise([x], w.t().contiguous(), sc, sh, ops.ACT_LEAKY, 0.2); b2 = ops.pointwise([x], w, sc, sh, ops.ACT_LEAKY, 0.2, w_rowmajor=True)
    tol = 1e-5 * max(1.0, ref.abs().max().item())
    chk("pointwise", (a.double() - ref).abs().max().item() < tol and (b2.double() - ref).abs().max().item() < tol, (B, K, Co, n))
for it in range(12):                                             # input-gradient weight pack == pack of the flipped, transposed filter
    Cout, Cin, taps = int(rs.choice([64, 128, 256])), rs.randint(1, 300), int(rs.choice([1, 9]))
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") if taps == 9 else torch.randn(Cout, Cin, device="cuda")
    want = ops.conv3x3_pack_weight(w.flip(2, 3).transpose(0, 1).contiguous()) if taps == 9 else ops.gemm_pack_weight(w.t().contiguous())
    chk("dgrad_pack", torch.equal(ops.conv_pack_weight_dgrad(w), want), (Cout, Cin, taps))
for it in range(12):                                             # kNN on cell lists (unorganised supports of >= 1024 points, K = 16) incl. clustered data
    B, S, Q = rs.randint(1, 4), int(rs.choice([1024, 1500, 2048, 5000, 9000])), rs.randint(1, 3000)
    sup = torch.rand(B, S, 3, device="cuda"); q = torch.rand(B, Q, 3, device="cuda")
    if it % 3 == 0: sup[:, : S // 2] = sup[:, : S // 2] * 0.01 + 0.5          # half of the support in one cell
    if it % 4 == 0: q = sup[:, torch.randint(0, S, (Q,), device="cuda")]       # queries ON support points (zero distances, ties)
    idx, d2 = ops.knn_batch(sup, q, 16, return_d2=True)
    dm = ((q[:, :, None, :] - sup[:, None, :, :]) ** 2).sum(-1)
    rv = torch.topk(dm, 16, dim=2, largest=False)[0]
    got = torch.gather(dm, 2, idx.long())
    chk("knn_cells", torch.allclose(got, rv, rtol=1e-5, atol=1e-7), (B, S, Q))
torch.cuda.synchronize()
print("fuzz done, mismatches:", bad)
