"""Two-rank rehearsal of the DDP + SyncBatchNorm training step on ONE GPU (gloo; both ranks on cuda:0): the fused SyncBatchNorm path, the
custom autograd functions under DistributedDataParallel, parameters identical on both ranks after the optimizer steps.  Launch:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 tools/ddp_rehearsal.py
Development aid (the 8-GPU RCCL run is the driver's)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist
from geometric_aware_dense_matching_amd import parallel, train_lm, synthetic
from geometric_aware_dense_matching_amd.config import make_model_cfg
from geometric_aware_dense_matching_amd.geoMatch import GeoMatch

rank, local_rank, world = parallel.init_distributed("gloo")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B, N, M = 2, 1024, 1024
torch.manual_seed(0)
model = GeoMatch(make_model_cfg(n_mesh_node=M, num_points=N), 1, model_points=synthetic.make_model_points(1, M)).to(dev)
ddp = parallel.wrap_for_training(model, local_rank=0)
n_sync = sum(isinstance(m, torch.nn.SyncBatchNorm) for m in ddp.modules())
opt = torch.optim.Adam(ddp.parameters(), lr=1e-4)
ds = train_lm.SyntheticCrops(B * world, N, M, seed=0)
batch = torch.utils.data.default_collate([ds[rank * B + i] for i in range(B)])
ddp.train()
losses = []
for step in range(3):
    out, _ = train_lm.model_fn_dec(ddp, batch, dev)
    out["loss"].backward()
    opt.step(); opt.zero_grad()
    losses.append(float(out["loss"].detach()))
torch.cuda.synchronize()
chk = torch.stack([p.detach().double().sum() for p in ddp.parameters()]).cpu()
rm = torch.stack([m.running_mean.double().sum() for m in ddp.modules() if isinstance(m, torch.nn.SyncBatchNorm)]).cpu()
lo, hi = chk.clone(), chk.clone()
dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
rlo, rhi = rm.clone(), rm.clone()
dist.all_reduce(rlo, op=dist.ReduceOp.MIN); dist.all_reduce(rhi, op=dist.ReduceOp.MAX)
ok = all(l == l and abs(l) < 1e6 for l in losses) and torch.equal(lo, hi) and torch.equal(rlo, rhi)
if rank == 0:
    print("ddp rehearsal: world=%d sync_bn_layers=%d losses=%s params_in_sync=%s running_stats_in_sync=%s -> %s" %
          (world, n_sync, ["%.4f" % l for l in losses], torch.equal(lo, hi), torch.equal(rlo, rhi), "OK" if ok else "FAIL"), flush=True)
parallel.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
