"""Inference drivers: the multi-object `test()` loop and the whole step as one hipGraph launch.

The reference's `test()` keeps one GeoMatch per object id and runs every detected instance as its own batch-1
forward (`cal_result_multimodel`, /root/reference/train_lm.py:298-314), recomputing the whole mesh branch per instance.
Here instances are grouped by object and each group is ONE batched pass (neighbour pyramid, forward, matching, pose);
results come back in the original instance order with the reference's concatenated layout
(`seg [bs,2,N]`, `rgbd [bs,128,N]`, `mesh [bs,128,M]`)."""
import torch

from . import matching, ops, pose, pyramid, settings

_PREC = {"bf16x3": ops.MATCH_BF16X3, "f32": ops.MATCH_F32, 0: 0, 1: 1}


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


def pipeline_step(model, inputs, precision="bf16x3", with_pose=False, keep_pyramid=False):
    """ONE pass of the hot path over a batch of crops resident on the device: neighbour pyramid (unless `inputs` already carries the
    loader's index arrays) -> GeoMatch.forward (eval) -> seg mask + descriptor packs + N x M arg-max (evaluator.py:78-93) [-> pose].
    Everything is enqueued on the current stream (and, with settings.USE_SIDE_STREAMS, on side streams forked from and joined back to
    it) with no host synchronisation, so the call captures in a hipGraph as it is.  Returns dict(seg, rgbd, mesh, mask, count,
    best_idx, best_sim[, RT, valid]) plus the 30 pyramid arrays when keep_pyramid."""
    prec = _PREC[precision]
    d = dict(inputs)
    pyr = None
    if "cld_nei_idx0" not in d:
        pyr = pyramid.build_pyramid(pyramid.cloud_view(d["cld_rgb_nrm"]), d["dpt_xyz"], overlap=True)
        d.update(pyr)
    ep = model(d, defer_seg=True)
    B, _, N = ep["rgbd"].shape
    M = ep["mesh"].shape[-1]
    mask, count, bi, bs = matching.match_tail(ep, B, N, M, prec)
    out = dict(seg=ep["seg"], rgbd=ep["rgbd"], mesh=ep["mesh"], mask=mask, count=count, best_idx=bi, best_sim=bs)
    if with_pose:
        out["RT"], out["valid"] = pose.solve_poses(out, d["cld_rgb_nrm"], model.model_emb.xyz)
    if keep_pyramid and pyr is not None:
        out.update((k, v) for k, v in pyr.items() if torch.is_tensor(v))
    return out


def outputs_equal(a, b):
    """Bit-for-bit comparison of two pipeline_step outputs (every tensor both hold).  Returns (all_equal, [names that differ])."""
    bad = [k for k in a if torch.is_tensor(a[k]) and k in b and not torch.equal(a[k], b[k])]
    return not bad, bad


class GraphedPipeline:
    """The whole step (pipeline_step) captured in a HIP graph and replayed from static input buffers: one host call per batch.

    TWO launch forms are captured and the better VALID one is what `__call__` replays (`self.form`):
      "single"  every kernel on one stream, in program order;
      "forked"  the same kernels with the neighbour pyramid, the mesh branch and the point branch on side streams
                (settings.USE_SIDE_STREAMS during the capture): parallel branches of the graph, ~12 % less time per step because the
                point branch's small kernels fill what the convolution launches leave of the chip.
    The forked form is only kept if its outputs are BIT-IDENTICAL to the single-stream EAGER step on the example inputs, after its
    first replays and again after a burst of back-to-back replays (`self.check` records every comparison); otherwise the pipeline
    falls back to the single-stream capture, and to nothing silently -- `self.form` and `self.check` say what runs.  `forked=False`
    captures the single-stream form only, `forked=True` requires the forked form (raises if it fails its check).

    All HIP operators of this package enqueue on torch's current stream with no host synchronisation and no allocation outside
    torch's graph-private pool, so they capture as they are; per-module caches (folded BN, packed weights, PReLU slopes) are filled
    by the eager warm-up passes before the capture.  Each form owns its scratch buffers (ops.BufferPool); the form that is not kept is
    freed after the decision (keep_both=True keeps both: `replay(form)`)."""

    def __init__(self, model, example_inputs, precision="bf16x3", with_pose=True, warmup=3, forked="auto", burst=8,
                 keep_pyramid=False, capture_error_mode=None, keep_both=False):
        self.model = model.eval()
        self.precision, self.with_pose, self.keep_pyramid = precision, with_pose, keep_pyramid
        self.static_in = {k: v.clone() for k, v in example_inputs.items() if torch.is_tensor(v)}
        self.graphs, self.outs, self.pools, self.check = {}, {}, {}, {}
        self._cap_kw = {"capture_error_mode": capture_error_mode} if capture_error_mode else {}
        caller_forked = bool(settings.USE_SIDE_STREAMS)     # the caller switched the forks on globally: capture as told, one form
        with torch.no_grad():
            if caller_forked:
                self._capture("forked", warmup)
                self.form = "forked"
            else:
                ref = self._eager_reference(warmup)
                self._capture("single", 1)
                self.check["single"] = self._compare(ref, "single", burst)
                self.form = "single"
                if forked in ("auto", True):
                    try:
                        settings.USE_SIDE_STREAMS = True
                        self._capture("forked", 2)
                        self.check["forked"] = self._compare(ref, "forked", burst)
                    except Exception as e:                              # noqa: BLE001 -- the forked form is a candidate only
                        torch.cuda.synchronize()
                        self.check["forked"] = {"bit_identical": False,
                                                "error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0] if str(e) else "")}
                        self.graphs.pop("forked", None)
                    finally:
                        settings.USE_SIDE_STREAMS = False
                    if self.check["forked"].get("bit_identical"):
                        self.form = "forked"
                    else:
                        self.graphs.pop("forked", None)
                        self.outs.pop("forked", None)
                        if forked is True:
                            raise RuntimeError("GraphedPipeline(forked=True): the forked capture is not bit-identical to the eager "
                                               "step: %r" % (self.check["forked"],))
        self.graph = self.graphs[self.form]
        self.static_out = self.outs[self.form]
        if not keep_both:                                            # the capture that lost keeps its private memory pool (activations of a
            for f in [f for f in self.graphs if f != self.form]:     # whole step): drop it unless the caller wants to time both (bench.py)
                del self.graphs[f], self.outs[f]
                self.pools.pop(f, None)

    # -- construction helpers ---------------------------------------------------------------------------------------------
    def _step(self):
        return pipeline_step(self.model, self.static_in, self.precision, self.with_pose, self.keep_pyramid)

    def _eager_reference(self, warmup):
        """Eager single-stream steps on the example inputs (they also fill the per-module caches); the last one's outputs, cloned."""
        pool = self.pools.setdefault("single", ops.BufferPool())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), ops.buffer_pool(pool):
            for _ in range(max(warmup, 1)):
                out = self._step()
            ref = {k: v.clone() for k, v in out.items() if torch.is_tensor(v)}
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        return ref

    def _capture(self, form, warmup):
        pool = self.pools.setdefault(form, ops.BufferPool())
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), ops.buffer_pool(pool):
            for _ in range(warmup):                                  # the form's scratch buffers (and side-stream allocations) exist
                self._step()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, **self._cap_kw), ops.buffer_pool(pool):
            out = self._step()
        self.graphs[form], self.outs[form] = g, out

    def _compare(self, ref, form, burst):
        g, out = self.graphs[form], self.outs[form]
        g.replay()
        g.replay()                                                   # twice: the second replay also reads what the first left behind
        torch.cuda.synchronize()
        ok1, bad1 = outputs_equal(ref, out)
        for _ in range(burst):                                       # back to back, no synchronisation in between
            g.replay()
        torch.cuda.synchronize()
        ok2, bad2 = outputs_equal(ref, out)
        return {"bit_identical": bool(ok1 and ok2), "first_replays_equal_eager": bool(ok1), "after_burst_equal_eager": bool(ok2),
                "burst": burst, "differing": sorted(set(bad1 + bad2))}

    # -- use --------------------------------------------------------------------------------------------------------------
    def __call__(self, inputs):
        """Copies `inputs` into the static buffers, replays the graph, returns the static outputs (valid until the next call)."""
        for k, buf in self.static_in.items():
            buf.copy_(inputs[k], non_blocking=True)
        self.graph.replay()
        return self.static_out

    def replay(self, form=None):
        """One replay of a captured form on the static inputs (default: the form __call__ uses); returns that form's static outputs."""
        form = form or self.form
        self.graphs[form].replay()
        return self.outs[form]
