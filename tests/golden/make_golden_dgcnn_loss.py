#!/usr/bin/env python3
"""dgcnn_losses.npz: the geoMatch_DGCNN variant's TRAINING matching loss (value and both gradients) from the REAL reference
(/root/reference/models/geoMatch_DGCNN.py:52-135, imported here as make_golden.py imports the rest; build container only).
The reference instance is the one make_golden.py builds for dgcnn_eval.npz (384 model vertices); positive_r is raised from 3 to 20
(mm per metre of depth) so that the positive sets are not just the ground-truth vertex itself."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG  # noqa: E402
import inputs as gin  # noqa: E402
from geometric_aware_dense_matching_amd import synthetic  # noqa: E402


def main():
    MG.install_reference()
    import models.geoMatch_DGCNN as GMD
    Md = 384
    mp = synthetic.make_model_points(1, Md)
    os.makedirs("/tmp/gdm_golden_kps", exist_ok=True)
    np.save("/tmp/gdm_golden_kps/obj_000001_fps.npy", mp)
    had = hasattr(np, "float")
    if not had:
        np.float = float                                               # dgcnn.py:190 uses the removed np.float alias
    dmodel = GMD.GeoMatch(dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, model_pth="/tmp/gdm_golden_kps", n_mesh_node=Md), 1)
    if not had:
        del np.float
    dmodel.positive_r = 20
    li = gin.dgcnn_loss_inputs(Md)
    rg = torch.from_numpy(li["rgbd_f"]).requires_grad_(True)
    mf = torch.from_numpy(li["mesh_f"]).requires_grad_(True)
    x = dict(origin_labels=torch.from_numpy(li["origin_labels"]), match_idx=torch.from_numpy(li["match_idx"]),
             visible_flag=torch.from_numpy(li["vis"]), RT=torch.from_numpy(li["RT"]))
    loss = dmodel.pointwise_feature_matching(rg, mf, x)
    loss.backward()
    mesh_xyz = dmodel.model_emb._buffers["mesh"][0][:3, :].transpose(0, 1).contiguous()
    np.savez_compressed(os.path.join(HERE, "dgcnn_losses.npz"), match_loss=loss.item(), rgbd_grad=rg.grad.numpy(), mesh_grad=mf.grad.numpy(),
                        mesh_xyz=mesh_xyz.numpy(), mesh_buffer=dmodel.model_emb._buffers["mesh"].numpy(), positive_r=np.float64(20))
    print("dgcnn_losses.npz: loss %.6f, |d rgbd| %.4e, |d mesh| %.4e" % (loss.item(), rg.grad.norm().item(), mf.grad.norm().item()))


if __name__ == "__main__":
    main()
