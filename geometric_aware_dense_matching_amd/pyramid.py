"""Neighbour pyramid on the GPU: the 22 exact-kNN searches per crop that the reference runs on the
CPU inside DataLoader workers (/root/reference/datasets/lm/linemod_pbr.py:515-569, the ycbv copy
ycbv_pbr.py:541-574), for a whole batch of crops, in TWO kernel launches (one per K class).

Input : cld  f32[B,N,3]   sampled scene points (first three rows of cld_rgb_nrm, transposed)
        dpt_xyz f32[B,S,S,3] the crop's per-pixel xyz map (dpt_xyz_clip)
Output: the 30 arrays the model consumes, batched, on the device; indices int32:
        cld_xyz{0-3} cld_nei_idx{0-3} cld_sub_idx{0-3} cld_interp_idx{0-3}
        r2p_ds_nei_idx{0-3} p2r_ds_nei_idx{0-3} r2p_up_nei_idx{0-2} p2r_up_nei_idx{0-2}

"Random" sub-sampling is a prefix slice of the pre-shuffled cloud (linemod_pbr.py:538), so all 22
searches depend only on the inputs and are independent of each other: they run concurrently.
"""
import torch

from . import ops

RGB_DS_SR = (4, 8, 8, 8)       # linemod_pbr.py:529
PCLD_SUB_SR = (4, 4, 4, 4)     # linemod_pbr.py:531
RGB_UP_SR = (4, 2, 2)          # linemod_pbr.py:556
K_NEI = 16


def cloud_from_inputs(cld_rgb_nrm):
    """cld_rgb_nrm f32[B,9,N] -> xyz f32[B,N,3] contiguous."""
    return cld_rgb_nrm[:, :3, :].transpose(1, 2).contiguous()


def cloud_view(cld_rgb_nrm):
    """cld_rgb_nrm f32[B,9,N] -> xyz f32[B,N,3] as a VIEW of the loader's tensor (no launch): build_pyramid makes it dense inside its
    own copy launch, together with the strided pixel grids and the prefix sub-clouds."""
    return cld_rgb_nrm[:, :3, :].transpose(1, 2)


READY = "_pyramid_ready"          # key of the event recorded behind an overlapped pyramid build
READY_CLOUD = "_pyramid_ready_cloud"   # ... and of the one behind its first part: the cloud's own K = 16 searches and pooling indices
KEEP = "_pyramid_keep"            # key of the buffers the build keeps alive with its results (kNN workspace, strided grids)
PYR_STREAM = 2                    # side-stream number of an overlapped build (its own: the point branch forks onto 0, the mesh branch onto 1)


def build_pyramid(cld, dpt_xyz, overlap=False, _keep=None):
    """overlap=True (inference, settings.USE_SIDE_STREAMS with "pyr" in SIDE_PARTS): the searches are enqueued on a side stream of
    their own, so that the image trunk's first stages -- which need no indices -- run beside them; the returned dict then carries
    the event EVERY consuming stream must wait for under READY (FFB6DEmb.forward does, on the main stream and on the point
    branch's fork; `wait_ready(pyr)` for other consumers).  Its tensors are allocated on that side stream; `wait_ready` also
    records them on the consuming stream, so their blocks are not reused while that stream still reads them."""
    if not (cld.is_cuda and dpt_xyz.is_cuda):
        raise RuntimeError("build_pyramid runs on the GPU (HIP kNN); there is no CPU fallback")
    from . import settings
    if overlap and settings.USE_SIDE_STREAMS and "pyr" in settings.SIDE_PARTS:
        with ops.fork(cld.device, PYR_STREAM) as f:
            f.use(cld, dpt_xyz)
            keep = []
            pyr = build_pyramid(cld, dpt_xyz, _keep=keep)
            ev = torch.cuda.Event()
            ev.record(f.side)
        pyr[READY] = ev                 # the non-tensor entries, and only of an overlapped build
        pyr[KEEP] = keep
        return pyr
    B, N, _ = cld.shape
    S = dpt_xyz.shape[1]
    assert dpt_xyz.shape == (B, S, S, 3)
    if N % 256 != 0 or N < 1024:
        raise ValueError("N=%d: four /4 levels must leave >= 16 points (N >= 1024, N %% 256 == 0)" % N)
    dpt_xyz = dpt_xyz.contiguous()
    n_lv = [N]
    for i in range(4):
        n_lv.append(n_lv[-1] // PCLD_SUB_SR[i])
    # the strided xyz maps (linemod_pbr.py:517-527) and the prefix sub-clouds (:538), dense, in ONE copy launch.  (The searches
    # could read the prefix views in place -- batch stride kept -- but every later consumer wants them dense.)  A cloud that arrives
    # as a view of the loader's channel-major tensor (cloud_view) is made dense by the same launch: [B,n,3,1] views, whose last
    # dimension is trivially contiguous.
    if cld.is_contiguous():
        cviews = [cld[:, :n_lv[i]] for i in (1, 2, 3, 4)]
    else:
        cviews = [cld[:, :n].unsqueeze(-1) for n in n_lv]
    dense = ops.copy_views([dpt_xyz[:, ::sc, ::sc, :] for sc in (2, 4, 8)] + cviews)
    grids = {1: dpt_xyz.reshape(B, S * S, 3)}
    for sc, g in zip((2, 4, 8), dense[:3]):
        grids[sc] = g.reshape(B, -1, 3)
    if cld.is_contiguous():
        levels = [cld] + dense[3:]
    else:
        levels = [t.squeeze(-1) for t in dense[3:]]
        dense = dense[:3] + levels

    jobs, names = [], []
    for i in range(4):
        cur, sub, px = levels[i], levels[i + 1], grids[RGB_DS_SR[i]]
        jobs += [(cur, cur, K_NEI), (sub, cur, 1), (px, sub, K_NEI, S // RGB_DS_SR[i]), (sub, px, 1)]     # px: an S/sr-wide pixel grid
        names += ["cld_nei_idx%d" % i, "cld_interp_idx%d" % i, "r2p_ds_nei_idx%d" % i, "p2r_ds_nei_idx%d" % i]
    for i in range(3):
        pts, px = levels[3 - i], grids[RGB_UP_SR[i]]
        jobs += [(px, pts, K_NEI, S // RGB_UP_SR[i]), (pts, px, 1)]
        names += ["r2p_up_nei_idx%d" % i, "p2r_up_nei_idx%d" % i]
    if _keep is not None:
        _keep += dense
    # Two launches of the job table: first what the point branch's first RandLA block needs -- the cloud's own K = 16 searches (one
    # knn_wave launch) and the pooling indices cut from them -- then the searches against and from the pixel grids and the
    # nearest-interpolation ones.  An overlapped build records an event behind the first part (READY_CLOUD): the point stream starts
    # ~170 us earlier than behind the whole pyramid, which is what the image stream waits for at the first fusion.
    first = [i for i, nm in enumerate(names) if nm.startswith("cld_nei_idx")]
    rest = [i for i in range(len(jobs)) if i not in first]
    outs = [None] * len(jobs)
    pyr = {}
    hold = _keep if _keep is not None else []          # both workspaces live until the function returns
    for i, o in zip(first, ops.knn_jobs([jobs[i] for i in first], B, keep_workspace=hold)):
        outs[i] = o
    pyr = dict((names[i], outs[i]) for i in first)
    subs = ops.copy_views([pyr["cld_nei_idx%d" % i][:, : n_lv[i + 1]] for i in range(4)])     # pooling indices: prefix rows, dense
    if _keep is not None:
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(cld.device))
        pyr[READY_CLOUD] = ev
    for i, o in zip(rest, ops.knn_jobs([jobs[i] for i in rest], B, keep_workspace=hold)):
        outs[i] = o
    pyr.update((names[i], outs[i]) for i in rest)
    pyr = dict([(nm, pyr[nm]) for nm in names] + [(k, v) for k, v in pyr.items() if k not in names])     # the reference's key order first
    for i in range(4):
        pyr["cld_xyz%d" % i] = levels[i]
        pyr["cld_sub_idx%d" % i] = subs[i]
    return pyr


def wait_ready(inputs, stream=None, cloud_only=False):
    """Make `stream` (default: the current one) wait for an overlapped pyramid build, if the inputs carry one, and record the
    pyramid's tensors on it.  Idempotent and cheap: every consuming stream calls it before its first index use.  cloud_only: wait
    only for the first part of the build (cld_nei_idx*, cld_sub_idx*, cld_xyz*: what a RandLA block reads); a later full wait is
    still needed before any other index is used."""
    ev = inputs.get(READY_CLOUD) if cloud_only else None
    if ev is None:
        ev = inputs.get(READY)
    if ev is None:
        return
    st = stream or torch.cuda.current_stream()
    st.wait_event(ev)
    for k, v in inputs.items():
        if torch.is_tensor(v) and v.is_cuda and (k.startswith("cld_") and k[4:7] in ("xyz", "nei", "sub", "int") or "_nei_idx" in k):
            v.record_stream(st)
    for v in inputs.get(KEEP, ()):
        v.record_stream(st)
