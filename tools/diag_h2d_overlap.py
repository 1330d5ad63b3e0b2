import sys, time, torch
sys.path.insert(0, '/root/repo')
from geometric_aware_dense_matching_amd import infer, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch
dev = torch.device("cuda", 0)
B, N, M = 16, 2048, 8192
torch.manual_seed(0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev).eval()
b = synthetic.make_batch(seed=1, batch=B, n_points=N)
d = {k: torch.from_numpy(b[k]).to(dev) for k in ("rgb", "cld_rgb_nrm", "choose", "dpt_xyz")}
host = {k: v.cpu().pin_memory() for k, v in d.items()}
stage = {k: torch.empty_like(v) for k, v in d.items()}
for forked in ("auto", False):
    gp = infer.GraphedPipeline(model, d, with_pose=False, forked=forked)
    cs = torch.cuda.Stream()
    def run(K, copy):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            if copy:
                with torch.cuda.stream(cs):
                    for k, v in host.items(): stage[k].copy_(v, non_blocking=True)
            gp.graph.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
    run(5, True)
    print("form", gp.form, "replay alone %.3f ms   with an independent H2D (26 MB) on another stream %.3f ms" % (run(20, False), run(20, True)))
    def copy_only(K):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(K):
            with torch.cuda.stream(cs):
                for k, v in host.items(): stage[k].copy_(v, non_blocking=True)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / K * 1e3
    print("   H2D alone %.3f ms" % copy_only(20))
# the pipelined form of bench.py's from_host leg: two staging sets, the next batch's copy on the copy stream beside the current replay
gp = infer.GraphedPipeline(model, d, with_pose=False)
stage2 = [{k: torch.empty_like(v) for k, v in d.items()} for _ in range(2)]
cs = torch.cuda.Stream(); main = torch.cuda.current_stream()
for K in (12, 40, 40):
    filled = [torch.cuda.Event() for _ in range(2)]; taken = [torch.cuda.Event() for _ in range(2)]
    def fill(i):
        with torch.cuda.stream(cs):
            cs.wait_event(taken[i % 2])
            for k, v in host.items(): stage2[i % 2][k].copy_(v, non_blocking=True)
            filled[i % 2].record(cs)
    for e in taken: e.record(main)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    fill(0)
    for i in range(K):
        if i + 1 < K: fill(i + 1)
        main.wait_event(filled[i % 2])
        for k, buf in gp.static_in.items(): buf.copy_(stage2[i % 2][k], non_blocking=True)
        taken[i % 2].record(main)                       # the staging set is free once it has been copied out
        gp.graph.replay()
    torch.cuda.synchronize()
    print("pipelined, K = %d: %.3f ms per step" % (K, (time.perf_counter() - t0) / K * 1e3))
