"""Multi-object inference driver.

The reference's `test()` keeps one GeoMatch per object id and runs every detected instance as its own batch-1
forward (`cal_result_multimodel`, /root/reference/train_lm.py:298-314), recomputing the whole mesh branch per instance.
Here instances are grouped by object and each group is ONE batched pass (neighbour pyramid, forward, matching, pose);
results come back in the original instance order with the reference's concatenated layout
(`seg [bs,2,N]`, `rgbd [bs,128,N]`, `mesh [bs,128,M]`)."""
import torch

from . import matching, pose, pyramid


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
            if "cld_nei_idx0" not in sub:
                sub.update(pyramid.build_pyramid(pyramid.cloud_from_inputs(sub["cld_rgb_nrm"]), sub["dpt_xyz"]))
            model = model_dict[cid]
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
