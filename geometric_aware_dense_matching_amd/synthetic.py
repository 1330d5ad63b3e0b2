"""Synthetic RGB-D crops and object models of the reference's shapes (host side, numpy).

There are no datasets here, so every test / bench input is generated.  The shapes and
conventions follow the reference loader; none of its code is used:

  * frame 480x640, LineMOD intrinsics (/root/reference/ref/lmo.py:87)
  * depth -> xyz as `dpt_2_pcld` (/root/reference/datasets/lm/linemod_pbr.py:398-411):
    x = (u - cx) * d / fx, y = (v - cy) * d / fy, z = d, zero where d == 0
  * square crop of side S=256 (config/lmo_cfg.py:98) -> `dpt_xyz_clip [S,S,3]`, rgb [3,S,S]
  * rgb normalised as `normalize_color` (/root/reference/utils/ply.py:502-509;
    note its std is [.229,.224,.224])
  * N valid pixels sampled without replacement and shuffled (linemod_pbr.py:476-496)
  * object model stored like `obj_%06d_fps.npy`: [M,9] = xyz(mm), rgb(uint8 range), normal
    (/root/reference/models/SplineCNN.py:180-193)

The depth is a smooth bumpy surface in 0.6-1.2 m plus per-pixel jitter, so all pairwise
distances are distinct (tie-free) unless `duplicates=True` is asked for, which reproduces
the loader's `np.pad(..., 'wrap')` duplicate points (linemod_pbr.py:492).
"""
import numpy as np

LM_K = np.array([[572.4114, 0.0, 325.2611], [0.0, 573.57043, 242.04899], [0.0, 0.0, 1.0]], dtype=np.float32)
COLOR_MEAN = np.array([0.485, 0.456, 0.406], dtype=np.float32)
COLOR_STD_CROP = np.array([0.229, 0.224, 0.224], dtype=np.float32)   # utils/ply.py:502
COLOR_STD_MESH = np.array([0.229, 0.224, 0.225], dtype=np.float32)   # utils/ply.py:522


def normalize_color(color, std=COLOR_STD_CROP):
    c = color.astype(np.float32) / 255.0
    c = c - COLOR_MEAN
    c = c / std
    return c.astype(np.float32)


def make_frame(rs, hole_frac=0.05):
    """One 480x640 RGB-D frame: depth (m) with zero holes, rgb uint8, unit normals."""
    H, W = 480, 640
    v, u = np.mgrid[:H, :W].astype(np.float32)
    ph = rs.rand(6).astype(np.float32) * 6.28
    depth = (0.9 + 0.12 * np.sin(u / 37.0 + ph[0]) * np.cos(v / 29.0 + ph[1])
             + 0.08 * np.sin(u / 11.0 + ph[2]) + 0.06 * np.cos(v / 7.0 + ph[3])
             + 0.03 * np.sin((u + v) / 5.0 + ph[4])).astype(np.float32)
    depth += (rs.rand(H, W).astype(np.float32) - 0.5) * 2e-3          # break exact ties
    depth = np.clip(depth, 0.6, 1.2).astype(np.float32)
    holes = rs.rand(H, W) < hole_frac
    depth[holes] = 0.0
    rgb = rs.randint(0, 256, size=(H, W, 3)).astype(np.uint8)
    nrm = rs.randn(H, W, 3).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=2, keepdims=True) + 1e-12
    return depth, rgb, nrm.astype(np.float32)


def depth_to_xyz(depth, K=LM_K):
    """`dpt_2_pcld` arithmetic (linemod_pbr.py:398-411): integer pixel maps minus a float32 intrinsic promote to
    float64, so the reference forms x, y in double from the float32 depth and rounds to float32 once, at the end."""
    H, W = depth.shape
    v, u = np.mgrid[:H, :W]
    d = depth.astype(np.float32)
    K = np.asarray(K, dtype=np.float32)
    msk = (d > 1e-8).astype(np.float32)
    x = (u - K[0][2]) * d / K[0][0]
    y = (v - K[1][2]) * d / K[1][1]
    xyz = np.stack([x, y, d.astype(np.float64)], axis=2) * msk[:, :, None]
    return xyz.astype(np.float32)


def make_crop(seed, n_points, S=256, duplicates=False):
    """One crop sample, host numpy, keys as the reference loader's item dict (model inputs
    only): rgb f32[3,S,S], cld_rgb_nrm f32[9,N], choose i32[1,N], dpt_xyz f32[S,S,3],
    labels i32[N]."""
    rs = np.random.RandomState(seed)
    depth, rgb, nrm = make_frame(rs)
    xyz = depth_to_xyz(depth)
    y0, x0 = (480 - S) // 2, (640 - S) // 2
    xyz_c = np.ascontiguousarray(xyz[y0:y0 + S, x0:x0 + S])
    rgb_c = normalize_color(rgb[y0:y0 + S, x0:x0 + S])
    nrm_c = np.ascontiguousarray(nrm[y0:y0 + S, x0:x0 + S])
    valid = (xyz_c[:, :, 2] > 1e-6).flatten().nonzero()[0].astype(np.uint32)
    if duplicates:
        keep = valid[rs.permutation(len(valid))[: (3 * n_points) // 4]]
        choose = np.pad(keep, (0, n_points - len(keep)), "wrap")          # linemod_pbr.py:492
    else:
        choose = valid[rs.permutation(len(valid))[:n_points]]
    choose = choose[rs.permutation(n_points)]
    cld = xyz_c.reshape(-1, 3)[choose]
    rgb_pt = rgb_c.reshape(-1, 3)[choose]
    nrm_pt = nrm_c.reshape(-1, 3)[choose]
    labels = (rs.rand(n_points) < 0.5).astype(np.int32)
    return dict(
        rgb=np.ascontiguousarray(rgb_c.transpose(2, 0, 1)).astype(np.float32),
        cld_rgb_nrm=np.ascontiguousarray(np.concatenate([cld, rgb_pt, nrm_pt], axis=1).T).astype(np.float32),
        choose=choose.astype(np.int32)[None, :],
        dpt_xyz=xyz_c.astype(np.float32),
        labels=labels,
    )


def make_batch(seed, batch, n_points, S=256, duplicates=False):
    items = [make_crop(seed * 1000 + i, n_points, S, duplicates) for i in range(batch)]
    return {k: np.stack([it[k] for it in items]) for k in items[0]}


def make_model_points(seed, n_vertices, diameter_mm=102.099):
    """Object model like obj_%06d_fps.npy: f32[M,9] = xyz (mm), rgb (0..255), unit normal.
    A noisy ellipsoid of the given diameter (obj_01: config/lmo_cfg.py:8)."""
    rs = np.random.RandomState(seed)
    d = rs.randn(n_vertices, 3)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    radii = np.array([0.5, 0.38, 0.3]) * diameter_mm
    xyz = d * radii * (1.0 + 0.03 * rs.randn(n_vertices, 1))
    nrm = d / radii
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    rgb = rs.randint(0, 256, size=(n_vertices, 3)).astype(np.float32)
    return np.concatenate([xyz, rgb, nrm], axis=1).astype(np.float32)


def synthetic_state_dict(template, seed=0):
    """Deterministic weights from (key name, shape, seed) so that any implementation with the
    reference's parameter names gets identical values without shipping a 94 MB checkpoint.
    `template`: mapping name -> tensor/array (only shapes and dtypes are read).
    BatchNorm running_var is kept positive, weights are scaled ~ kaiming so activations stay O(1)."""
    import hashlib
    import re
    import torch
    out = {}
    for name, t in template.items():
        shape = tuple(t.shape)
        # `final` is ONE module registered under two names (models/ffb6d.py:79-80): a real checkpoint
        # holds identical tensors under both, so both names must hash alike.
        canon = name.replace("cnn_up_stages.2.0.", "cnn_up_stages.3.1.")
        # DGCNN registers each BatchNorm twice: as `bnN` and inside `convN = Sequential(conv, bnN, act)`
        # (models/dgcnn.py:67-80), i.e. keys bnN.* and convN.1.* alias one tensor.
        canon = re.sub(r"(^|\.)conv(\d)\.1\.", r"\1bn\2.", canon)
        h = int.from_bytes(hashlib.sha256((canon + "|%d" % seed).encode()).digest()[:4], "little")
        rs = np.random.RandomState(h)
        if name.endswith("num_batches_tracked"):
            out[name] = torch.tensor(1, dtype=torch.long)
            continue
        if not torch.is_floating_point(torch.as_tensor(t)):
            out[name] = torch.as_tensor(t).clone()
            continue
        if name.endswith("running_var"):
            a = 0.5 + rs.rand(*shape)
        elif name.endswith("running_mean"):
            a = 0.1 * rs.randn(*shape)
        elif name.endswith("bias"):
            a = 0.05 * rs.randn(*shape)
        elif len(shape) == 1 and shape[0] == 1:
            a = 0.25 + 0.05 * rs.rand(*shape)                 # PReLU slope
        elif len(shape) == 1:
            a = 1.0 + 0.1 * rs.randn(*shape)                  # BN gamma / awl params
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            a = rs.randn(*shape) * np.sqrt(1.0 / max(fan_in, 1))
        out[name] = torch.from_numpy(np.asarray(a, dtype=np.float32).reshape(shape))
    return out
