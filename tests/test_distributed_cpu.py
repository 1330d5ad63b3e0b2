"""CPU rehearsal of the N>1 path with world_size-2 gloo (no GPU): shard arithmetic, the bench's max-over-ranks
timing rule, and DDP gradient averaging over the product's pure-torch head / loss modules."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    from geometric_aware_dense_matching_amd import parallel
    from geometric_aware_dense_matching_amd.layers import PtSeq
    from geometric_aware_dense_matching_amd.loss import CircleLoss, FocalLoss
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    # 1. shards: disjoint, covering, balanced
    lo, hi = parallel.shard_range(37, rank, world)
    sizes = [torch.zeros(1, dtype=torch.long) for _ in range(world)]
    dist.all_gather(sizes, torch.tensor([hi - lo]))
    assert sum(int(s) for s in sizes) == 37 and max(int(s) for s in sizes) - min(int(s) for s in sizes) <= 1
    # 2. timing rule
    assert parallel.max_over_ranks(1.0 + rank) == float(world)
    assert parallel.sum_over_ranks(1.0) == float(world)
    # 3. DDP over the head + losses: averaged gradient == gradient of the mean loss over both shards
    torch.manual_seed(0)
    head = PtSeq(16).conv1d(16, bn=False).conv1d(2, activation=None)
    ddp = parallel.wrap_for_training(head, sync_bn=False)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 16, 50, generator=g)
    y = torch.randint(0, 2, (4, 50), generator=g)
    sim = torch.rand(4, 20, 30, generator=g) * 2 - 1
    msk = torch.rand(4, 20, 30, generator=g) < 0.2
    msk[:, :, 0] = True
    focal, circle = FocalLoss(gamma=2), CircleLoss(16)
    scale = torch.nn.Parameter(torch.ones(()))
    lo, hi = parallel.shard_range(4, rank, world)
    loss = focal(ddp(x[lo:hi]), y[lo:hi]) + sum(circle(sim[i] * scale, msk[i], 0.2) for i in range(lo, hi)) / (hi - lo)
    loss.backward()
    grads = [p.grad.clone() for p in head.parameters()]
    if rank == 0:
        torch.manual_seed(0)
        ref = PtSeq(16).conv1d(16, bn=False).conv1d(2, activation=None)
        full = 0.5 * (focal(ref(x[:2]), y[:2]) + focal(ref(x[2:]), y[2:]))
        full.backward()
        for a, b in zip(grads, [p.grad for p in ref.parameters()]):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7)
        out.put("ok")
    parallel.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get() == "ok"


def test_shard_range_edge_cases():
    from geometric_aware_dense_matching_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 9, 100):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_train_cli_flags_match_the_reference():
    from geometric_aware_dense_matching_amd import train_lm
    a = train_lm.build_parser().parse_args("--gpus=2 -state=train -dataset_name=lmo -cls_id=1 -checkpoint=ck/".split())
    assert (a.gpus, a.state, a.dataset_name, a.cls_id, a.checkpoint) == (2, "train", "lmo", 1, "ck/")
    a = train_lm.build_parser().parse_args("--gpus=0 -state=test".split())
    assert a.state == "test" and a.cls_id == 5 and a.bn_momentum == 0.9 and a.decay_step == 2e5


def test_bench_gpus_flag_starts_the_ranks_itself():
    """`python bench.py --gpus 2` with no launcher around it (the flag was parsed and never read): the parent starts two
    fresh ranks through torch.distributed.run BEFORE importing torch (nothing in the parent touches the GPU), rank 0's JSON line
    comes back on the parent's stdout, the exit code is the children's.  --probe-ranks keeps it to the rendezvous (gloo, no GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--probe-ranks"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    rec = json.loads(lines[0])
    assert rec["n_ranks_seen"] == 2 and rec["world_size"] == 2 and rec["spawned_by_bench"] and not rec["torch_cuda_initialised"]
    # the parent: launch_ranks() builds the reference-shaped launch line and has not imported torch when it runs it
    code = ("import sys, json; sys.argv = ['bench.py']; sys.path.insert(0, %r); import bench\n"
            "seen = {}\n"
            "class R:\n    returncode = 7\n"
            "def run(cmd, env):\n    seen['cmd'] = cmd; seen['torch'] = 'torch' in sys.modules; seen['env'] = env.get('GDM_BENCH_SPAWNED'); return R()\n"
            "argv = ['--gpus', '4', '--steps', '3']\n"
            "rc = bench.launch_ranks(bench.parse(argv), argv, run=run)\n"
            "print(json.dumps(dict(seen, rc=rc)))\n" % root)
    q = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert q.returncode == 0, q.stderr[-2000:]
    seen = json.loads(q.stdout.strip().splitlines()[-1])
    assert seen["rc"] == 7 and seen["torch"] is False and seen["env"] == "1"
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]


def test_exact_f32_flag_covers_every_split_bf16_switch():
    """bench.py --exact-f32 zeroes the environment variable of EVERY switch in settings.SPLIT_BF16_SWITCHES (a hard-coded
    list had missed USE_OWN_STEM)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    from geometric_aware_dense_matching_amd import settings
    envs = bench.exact_f32_env()
    assert len(envs) == len(settings.SPLIT_BF16_SWITCHES) and "GDM_OWN_STEM" in envs
    for name, env in zip(settings.SPLIT_BF16_SWITCHES, envs):
        assert env == "GDM_" + name[len("USE_"):], (name, env)
