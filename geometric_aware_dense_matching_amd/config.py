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


YCBV_DIAMETERS = {1: 172.063, 2: 269.573, 3: 198.377, 4: 120.543, 5: 196.463, 6: 89.797, 7: 142.543, 8: 114.053, 9: 129.540,
                  10: 197.796, 11: 259.534, 12: 259.566, 13: 161.922, 14: 124.990, 15: 226.170, 16: 237.299, 17: 203.973,
                  18: 121.365, 19: 174.746, 20: 217.094, 21: 102.903}        # config/ycbv_cfg.py:3-25

LMO_OBJS = {1: "ape", 5: "can", 6: "cat", 8: "driller", 9: "duck", 10: "eggbox", 11: "glue", 12: "holepuncher"}   # lmo_cfg.py:74-90
YCBV_OBJS = {1: "002_master_chef_can", 2: "003_cracker_box", 3: "004_sugar_box", 4: "005_tomato_soup_can", 5: "006_mustard_bottle",
             6: "007_tuna_fish_can", 7: "008_pudding_box", 8: "009_gelatin_box", 9: "010_potted_meat_can", 10: "011_banana",
             11: "019_pitcher_base", 12: "021_bleach_cleanser", 13: "024_bowl", 14: "025_mug", 15: "035_power_drill",
             16: "036_wood_block", 17: "037_scissors", 18: "040_large_marker", 19: "051_large_clamp",
             20: "052_extra_large_clamp", 21: "061_foam_brick"}                                                       # ycbv_cfg.py:75-97

# What `-dataset_name` selects (the reference imports config/<name>_cfg.py at module top: train_lm.py:17, train_ycb.py:18).
# Only the keys the hot path and the entry points read.  (config/lmfull_cfg.py is stale in the reference: its MODEL dict lacks
# model_d / neighbor_dis_th / model_name, no entry point imports it.)
DATASET_CONFIGS = {
    "lmo": dict(model_name="lmo", diameters=LM_DIAMETERS, objs=LMO_OBJS, sym_objs=("eggbox",), neighbor_dis_th=0.02,
                model_pth="datasets/lm/linemod/kps", checkpoints="train_log/lm/checkpoints", train_batch_size=24,
                val_batch_size=128, n_points=4096, n_mesh=4096, load_strict=True),                    # config/lmo_cfg.py:95-133
    "ycbv": dict(model_name="ycbv", diameters=YCBV_DIAMETERS, objs=YCBV_OBJS,
                 sym_objs=("024_bowl", "052_extra_large_clamp", "061_foam_brick"), neighbor_dis_th=0.06,
                 model_pth="datasets/ycbv/ycbv/kps", checkpoints="train_log/ycb/checkpoints", train_batch_size=8,
                 val_batch_size=128, n_points=4096, n_mesh=4096, load_strict=False),                  # config/ycbv_cfg.py:100-136; train_ycb.py:140
}


def dataset_config(name):
    if name not in DATASET_CONFIGS:
        raise KeyError("-dataset_name=%r: known datasets are %s" % (name, sorted(DATASET_CONFIGS)))
    return DATASET_CONFIGS[name]


def make_model_cfg(n_mesh_node=4096, num_points=4096, model_pth=None, model_name=None, model_d=None, dataset="lmo",
                   neighbor_dis_th=None):
    """The `MODEL` dict of config/<dataset>_cfg.py (lmo_cfg.py:121-133, ycbv_cfg.py:124-136)."""
    ds = dataset_config(dataset)
    return dict(n_mesh_node=n_mesh_node, feat_dim=128, checkpoints=ds["checkpoints"],
                model_pth=ds["model_pth"] if model_pth is None else model_pth,
                ffb_config=ConfigRandLA(num_points), resnet_dir="models/cnn/ResNet_pretrained_mdl",
                model_d=dict(ds["diameters"] if model_d is None else model_d),
                neighbor_dis_th=ds["neighbor_dis_th"] if neighbor_dis_th is None else neighbor_dis_th,
                model_name=ds["model_name"] if model_name is None else model_name)


def make_dgcnn_cfg(n_mesh_node=4096, dataset="ycbv", model_pth=None, model_d=None):
    """cfg dict of models/geoMatch_DGCNN.py:13-36 (feat_dim, k, embed_dim, dropout, model_pth, n_mesh_node) plus the dataset's
    diameters / radius factor that its matching loss reads (geoMatch_DGCNN.py:66-67)."""
    ds = dataset_config(dataset)
    return dict(feat_dim=128, k=16, embed_dim=1024, dropout=0.1, n_mesh_node=n_mesh_node,
                model_pth=ds["model_pth"] if model_pth is None else model_pth,
                model_d=dict(ds["diameters"] if model_d is None else model_d), neighbor_dis_th=ds["neighbor_dis_th"],
                model_name=ds["model_name"])
