"""GPU-side front end: from a device-resident RGB-D frame and a detection box to the model's input dict, so that the
DataLoader only ships the frame.  Mirrors the geometric part of the reference loader
(/root/reference/datasets/lm/linemod_pbr.py): depth -> xyz (`dpt_2_pcld` :398-411), the S x S crop (:468-473; integer crops
only -- the reference's warpAffine resampling with a zoom is host-side preprocessing and stays out of scope), valid-pixel
sampling of N points with wrap-around padding (:476-496), `cld_rgb_nrm` / `choose` assembly (:498-513) and the neighbour
pyramid (:515-569)."""
import torch

from . import _lib, ops, pyramid
from ._lib import check


def depth_to_xyz(depth, K, origin, S):
    """depth f32[B,H,W] (m), K f32[B,3,3], origin i32[B,2] = (x0,y0) -> xyz f32[B,S,S,3]."""
    depth = ops._dev(depth, torch.float32, "depth")
    K = ops._dev(K, torch.float32, "K")
    origin = ops._idx32(origin, "origin")
    B, H, W = depth.shape
    out = torch.empty((B, S, S, 3), dtype=torch.float32, device=depth.device)
    check(_lib.lib().gdm_depth_to_xyz_hip(depth.data_ptr(), K.data_ptr(), origin.data_ptr(), B, H, W, S, out.data_ptr(), ops._stream()),
          "gdm_depth_to_xyz_hip")
    return out


def sample_valid_pixels(xyz, n_points, generator=None):
    """Choose n_points valid pixels (z > 1e-6) per crop uniformly without replacement, in random order; crops with fewer valid
    pixels wrap around (np.pad(..., 'wrap'), linemod_pbr.py:492).  xyz f32[B,S,S,3] -> choose i32[B,1,N]."""
    B, S = xyz.shape[0], xyz.shape[1]
    valid = xyz[..., 2].reshape(B, S * S) > 1e-6
    key = torch.rand((B, S * S), device=xyz.device, generator=generator)
    key = torch.where(valid, key, key + 2.0)                       # invalid pixels sort last
    order = torch.argsort(key, dim=1)                              # random permutation of the valid pixels first
    nvalid = valid.sum(dim=1, keepdim=True).clamp(min=1)
    j = torch.arange(n_points, device=xyz.device).unsqueeze(0) % nvalid      # wrap-around padding
    return torch.gather(order, 1, j).to(torch.int32).unsqueeze(1)


def make_inputs(rgb_norm, depth, normals, K, origin, S, n_points, generator=None):
    """rgb_norm f32[B,3,H,W] (already colour-normalised), depth f32[B,H,W], normals f32[B,3,H,W], K f32[B,3,3],
    origin i32[B,2] -> the model's input dict incl. the neighbour pyramid, all on the device."""
    B = depth.shape[0]
    xyz = depth_to_xyz(depth, K, origin, S)                                          # [B,S,S,3]
    ys = (origin[:, 1:2].long() + torch.arange(S, device=depth.device)[None]).clamp(0, depth.shape[1] - 1)
    xs = (origin[:, 0:1].long() + torch.arange(S, device=depth.device)[None]).clamp(0, depth.shape[2] - 1)
    bidx = torch.arange(B, device=depth.device)[:, None, None]
    rgb_c = rgb_norm[bidx, :, ys[:, :, None], xs[:, None, :]].permute(0, 3, 1, 2).contiguous()       # [B,3,S,S]
    nrm_c = normals[bidx, :, ys[:, :, None], xs[:, None, :]].permute(0, 3, 1, 2).contiguous()
    choose = sample_valid_pixels(xyz, n_points, generator)                           # [B,1,N]
    ch = choose[:, 0].long()
    cld = torch.gather(xyz.reshape(B, S * S, 3), 1, ch[:, :, None].expand(-1, -1, 3))
    rgb_pt = torch.gather(rgb_c.reshape(B, 3, S * S), 2, ch[:, None, :].expand(-1, 3, -1))
    nrm_pt = torch.gather(nrm_c.reshape(B, 3, S * S), 2, ch[:, None, :].expand(-1, 3, -1))
    inputs = dict(rgb=rgb_c, cld_rgb_nrm=torch.cat([cld.transpose(1, 2), rgb_pt, nrm_pt], dim=1).contiguous(), choose=choose,
                  dpt_xyz=xyz)
    inputs.update(pyramid.build_pyramid(cld.contiguous(), xyz))
    return inputs
