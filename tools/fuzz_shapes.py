"""Random small / odd shapes through the main operators against torch references (development aid: looks for latent faults)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from geometric_aware_dense_matching_amd import ops, randla, settings
rs = np.random.RandomState(0)
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
torch.cuda.synchronize()
print("fuzz done, mismatches:", bad)
