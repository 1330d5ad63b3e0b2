"""Multi-object inference driver.

The reference's `test()` keeps one GeoMatch per object id and runs every detected instance as its own batch-1
forward (`cal_result_multimodel`, /root/reference/train_lm.py:298-314), recomputing the whole mesh branch per instance.
Here instances are grouped by object and each group is ONE batched pass (neighbour pyramid, forward, matching, pose);
results come back in the original instance order with the reference's concatenated layout
(`seg [bs,2,N]`, `rgbd [bs,128,N]`, `mesh [bs,128,M]`)."""
import torch

from . import matching, ops, pose, pyramid


def run_multi_object(model_dict, inputs, cls_ids, with_pose=True, precision="bf16x3"):
    """model_dict: {cls_id: GeoMatch (eval, on the GPU)}; inputs: dict of batched device tensors (loader keys, plus
    `dpt_xyz` when the neighbour pyramid is not already in it); cls_ids: int tensor/list [bs].
    Returns dict(seg, rgbd, mesh, mask, best_idx, best_sim[, RT, valid])."""
    cls = torch.as_tensor(cls_ids).cpu().tolist()
    bs = len(cls)
    out = {}
    order = []
    with torch.no_grad():
        for cid in sorted(set(cls)):
            sel = [i for i, c in enumerate(cls) if c == cid]
            idx = torch.tensor(sel, device=inputs["cld_rgb_nrm"].device)
            sub = {k: v.index_select(0, idx) for k, v in inputs.items() if torch.is_tensor(v) and v.shape[:1] == (bs,)}
            model = model_dict[cid]
            if getattr(model, "needs_pyramid", True) and "cld_nei_idx0" not in sub:
                sub.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(sub["cld_rgb_nrm"]), sub["dpt_xyz"]))
            ep = model(sub)
            res = matching.match_frames(ep, precision=precision)
            part = dict(seg=ep["seg"], rgbd=ep["rgbd"], mesh=ep["mesh"].expand(len(sel), -1, -1), mask=res["mask"],
                        best_idx=res["best_idx"], best_sim=res["best_sim"])
            if with_pose:
                part["RT"], part["valid"] = pose.solve_poses(res, sub["cld_rgb_nrm"], model.model_emb.xyz)
            for k, v in part.items():
                out.setdefault(k, []).append(v)
            order += sel
    inv = torch.empty(bs, dtype=torch.long)
    inv[torch.tensor(order)] = torch.arange(bs)
    inv = inv.to(inputs["cld_rgb_nrm"].device)
    return {k: torch.cat(v, dim=0).index_select(0, inv) for k, v in out.items()}


class GraphedPipeline:
    """The whole step -- neighbour pyramid, GeoMatch.forward (eval), matching, pose -- captured ONCE in a HIP graph and
    replayed from static input buffers.  At batch 1 the eager step is launch-bound (~300 kernel launches, 6.6 ms of host time
    for ~2 ms of GPU work); a replay is a single launch.  All HIP operators of this package enqueue on torch's current stream
    with no host synchronisation and no allocation outside torch's graph-private pool, so they capture as they are; per-module
    caches (folded BN, packed weights, PReLU slopes) are filled by the eager warm-up passes before the capture."""

    def __init__(self, model, example_inputs, precision="bf16x3", with_pose=True, warmup=3):
        self.model = model.eval()
        self.precision, self.with_pose = precision, with_pose
        self.static_in = {k: v.clone() for k, v in example_inputs.items() if torch.is_tensor(v)}
        self.pool = ops.BufferPool()                                 # this graph's scratch buffers (and their captured zero fills)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad(), ops.buffer_pool(self.pool):
            for _ in range(warmup):                                  # per-module caches fill
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph), ops.buffer_pool(self.pool):
            self.static_out = self._step()

    def _step(self):
        d = dict(self.static_in)
        if "cld_nei_idx0" not in d:
            d.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(d["cld_rgb_nrm"]), d["dpt_xyz"], overlap=True))
        ep = self.model(d)
        res = matching.match_frames(ep, precision=self.precision)
        out = dict(seg=ep["seg"], rgbd=ep["rgbd"], mesh=ep["mesh"], mask=res["mask"], count=res["count"],
                   best_idx=res["best_idx"], best_sim=res["best_sim"])
        if self.with_pose:
            out["RT"], out["valid"] = pose.solve_poses(res, d["cld_rgb_nrm"], self.model.model_emb.xyz)
        return out

    def __call__(self, inputs):
        """Copies `inputs` into the static buffers, replays the graph, returns the static outputs (valid until the next call)."""
        for k, buf in self.static_in.items():
            buf.copy_(inputs[k], non_blocking=True)
        self.graph.replay()
        return self.static_out
