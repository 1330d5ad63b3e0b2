"""Launches the step's dominant kernels a few times each at the shapes the step uses, for `rocprofv3 --pmc` passes:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out_fetch -- python3 tools/pmc_kernels.py
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out_write -- python3 tools/pmc_kernels.py
    python tools/pmc_kernels.py --summarize out_fetch/*/*counter_collection.csv out_write/*/*counter_collection.csv > profiles/r04_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are KiB per dispatch; on gfx950 FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md,
HBM section), so it is doubled; other access widths are uncalibrated (noted per kernel in the output)."""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

B, N, M = 16, 2048, 8192
# kernel-name substring -> (what, algorithmic bytes per launch)
KERNELS = collections.OrderedDict([
    ("match_pipe_sim_kernel", ("N x 8192 descriptor kernel, materialised", 4.0 * 128 * (B * N + M) + 4.0 * B * N * M)),
    ("match_pipe_kernel", ("N x 8192 descriptor kernel, fused arg-max", 4.0 * 128 * (B * N + M) + 8.0 * B * N)),
    ("conv_mfma16_kernel<0, false, 9, false, 8, 8", ("3x3 512->512 @32x32 (packed operands in, fp32 NCHW out)",
                                                       B * 34 * 34 * 4 * 512.0 + 9 * 4 * 512 * 512.0 + 4.0 * B * 512 * 1024)),
    ("conv_mfma16_kernel<0, false, 1, false, 8, 9", ("1x1 1024->2304 @32x32 (tap GEMM of up_1, 144-channel tiles)",
                                                       B * 34 * 34 * 8 * 512.0 + 8 * 2304 * 512.0 + 4.0 * B * 2304 * 1024)),
    ("knn_cells_kernel", ("the cloud's own K = 16 search (2048 x 2048 per crop), cell lists", None)),
    ("knn_wave_kernel", ("K = 16 searches of the pyramid (small unorganised supports)", None)),
    ("knn_kernel<1>", ("K = 1 searches of the pyramid", None)),
    ("knn_grid_kernel", ("K = 16 searches against organised supports (window search)", None)),
    ("gather_max_", ("gather + max over K, C=64, 16384 px -> 512 points", 4.0 * B * (64 * 16384 + 16 * 512 + 64 * 512))),
    ("stem_kernel", ("conv7x7/2 + BN + ReLU + max-pool, 256^2 -> 64 x 64^2", 4.0 * B * 3 * 65536 + 49152 + 4.0 * B * 64 * 4096 * 2)),
    ("lfa_stage_mfma_kernel<32", ("LFA stage level 0 (n = 2048, D = 32), MFMA form", None)),
    ("upconv_final_points_kernel", ("up_3 + final at the 2048 sampled pixels of 256^2", 4.0 * B * 64 * (128 * 128 + N))),
])


def launch():
    import torch
    from geometric_aware_dense_matching_amd import _lib, ops, pyramid, synthetic
    L = _lib.lib()
    dev = torch.device("cuda")
    torch.manual_seed(0)
    scene, model = torch.randn(B, 128, N, device=dev), torch.randn(128, M, device=dev)
    sim = torch.empty(B, N, M, device=dev)
    srows, mrows = ops.match_pack(scene, 0), ops.match_pack(model, 0)
    x = torch.randn(B, 512, 32, 32, device=dev)
    wpk = ops.conv3x3_pack_weight(torch.randn(512, 512, 3, 3, device=dev) * 0.02)
    xpk = ops.conv3x3_pack_act(x)
    x1 = torch.randn(B, 1024, 1024, device=dev)
    w1 = ops.gemm_pack_weight(torch.randn(2304, 1024, device=dev) / 32)
    batch = synthetic.make_batch(seed=100, batch=B, n_points=N)
    cld = pyramid.cloud_from_inputs(torch.from_numpy(batch["cld_rgb_nrm"]).to(dev))
    xyz = torch.from_numpy(batch["dpt_xyz"]).to(dev)
    feat = torch.randn(B, 64, 128 * 128, device=dev)
    idx = torch.randint(0, 128 * 128, (B, N // 4, 16), device=dev, dtype=torch.int32)
    rgb = torch.randn(B, 3, 256, 256, device=dev)
    swp = ops.stem_pack_weight(torch.randn(64, 3, 7, 7, device=dev) * 0.1)
    sc, sh = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    pyr = pyramid.build_pyramid(cld, xyz)
    D = 32
    w = {k: torch.randn(*s, device=dev) * 0.1 for k, s in dict(w1t=(10, D // 2), wf=(D, D), wm=(D, D // 2)).items()}
    s1 = torch.ones(D // 2, device=dev)
    f0 = torch.randn(B, D // 2, N, device=dev)
    xs = torch.randn(B, 128 * 128, 64, device=dev)
    ch = torch.randint(0, 256 * 256, (B, N), device=dev, dtype=torch.int32)
    wk = ops.upconv_fused64_pack_weight(torch.randn(64, 64, 3, 3, device=dev) * 0.05)
    wf = ops.pack_rows64(torch.randn(64, 64, device=dev) / 8)
    bf = torch.zeros(64, device=dev)
    for _ in range(5):
        ops.match_packed(srows, mrows, B, N, M, 0, return_sim=True, sim_out=sim)
        ops.match_packed(srows, mrows, B, N, M, 0)
        ops.conv3x3_bf16x3(xpk, wpk, 512)
        ops.gemm_bf16x3(x1, w1, 2304)
        pyramid.build_pyramid(cld, xyz)
        ops.gather_max(feat, idx)
        ops.stem(rgb, swp, sc, sh)
        ops.lfa_stage(pyr["cld_xyz0"], pyr["cld_nei_idx0"], f0, w["w1t"], s1, s1, None, None, None, w["wf"], w["wm"], s1, s1)
        ops.upconv_final_points(xs, (128, 128), ch, wk, sc, sh, 2, 0.25, wf, bf, (256, 256))
    torch.cuda.synchronize()
    print("ok")


def summarize(fetch_csv, write_csv):
    def per_kernel(path, counter):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return acc
    fe, wr = per_kernel(fetch_csv, "FETCH_SIZE"), per_kernel(write_csv, "WRITE_SIZE")
    out = {"_comment": "HBM-side traffic per launch from rocprofv3 PMC passes (tools/pmc_kernels.py, B=16 N=2048 M=8192): FETCH_SIZE (KiB, "
                       "doubled: gfx950 counts half the bytes of wide coalesced reads) + WRITE_SIZE (KiB).  Calibrated for 16-byte-per-lane "
                       "streaming accesses; kernels with narrower accesses are marked uncalibrated.", "kernels": {}}
    for pat, (what, alg) in KERNELS.items():
        f = [v for k, vs in fe.items() if pat in k for v in vs]
        w = [v for k, vs in wr.items() if pat in k for v in vs]
        if not f or not w:
            continue
        fk, wk = sum(f) / len(f), sum(w) / len(w)
        out["kernels"][pat] = {"what": what, "fetch_size_kib_raw": round(fk, 1), "write_size_kib": round(wk, 1),
                               "bytes_per_launch": int(round((2 * fk + wk) * 1024)), "algorithmic_bytes_per_launch": alg,
                               "dispatches": [len(f), len(w)]}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--summarize":
        summarize(sys.argv[2], sys.argv[3])
    else:
        launch()
