"""Configuration objects with the reference's key names (config/lmo_cfg.py:95-133, common.py:12-27).
Only the keys the hot path reads are present; dataset / solver sections are out of scope."""

LM_DIAMETERS = {1: 102.099, 2: 247.506, 3: 167.355, 4: 172.492, 5: 201.404, 6: 154.546, 7: 124.264, 8: 261.472,
                9: 108.999, 10: 164.628, 11: 175.889, 12: 145.543, 13: 278.078, 14: 282.601, 15: 212.358}


class ConfigRandLA:
    k_n = 16
    num_layers = 4
    in_c = 9
    sub_sampling_ratio = [4, 4, 4, 4]
    d_out = [32, 64, 128, 256]

    def __init__(self, num_points=4096):
        self.num_points = num_points
        self.num_sub_points = [num_points // 4, num_points // 16, num_points // 64, num_points // 256]


def make_model_cfg(n_mesh_node=4096, num_points=4096, model_pth="datasets/lm/linemod/kps", model_name="lmo",
                   model_d=None):
    """The `MODEL` dict of config/lmo_cfg.py:121-133."""
    return dict(n_mesh_node=n_mesh_node, feat_dim=128, checkpoints="train_log/lm/checkpoints", model_pth=model_pth,
                ffb_config=ConfigRandLA(num_points), resnet_dir="models/cnn/ResNet_pretrained_mdl",
                model_d=dict(LM_DIAMETERS if model_d is None else model_d), neighbor_dis_th=0.02, model_name=model_name)
