"""Config -- same JSON schema and attribute names as the reference's cfg.Config (cfg.py:8-114).

`training_strategy` accepts the reference's "forloop" / "vmap" and additionally "hip" (the fused
MI355X iteration).  Keys the reference never reads (color_scaling, opacity_scaling,
hidden_layers_block, epochs, pose_lr) stay unread here too.
"""
import json
import os

import numpy as np


def load_matrix_from_txt(path, shape=(4, 4)):
    with open(path) as f:
        txt = f.readlines()
    txt = "".join(txt).replace("\n", " ")
    return np.array([float(v) for v in txt.split()]).reshape(shape)


# The schema as DATA: attribute -> (section, key[, conversion]).  Attribute names and JSON keys are the reference's
# (cfg.py:16-114: they are the interface the rest of objnerf/ reads); a missing key raises KeyError exactly as the
# reference's chain of subscripts does.  Derived attributes follow in Config.__init__.
_PLAIN = {
    # trainer / dataset                                                        cfg.py:16-31
    "start": ("trainer", "start"), "stride": ("trainer", "stride"), "do_bg": ("trainer", "do_bg", bool),
    "training_device": ("trainer", "train_device"), "data_device": ("trainer", "data_device"),
    "max_n_models": ("trainer", "n_models"), "imap_mode": ("trainer", "imap_mode"),
    "training_strategy": ("trainer", "training_strategy"),          # "forloop" "vmap" "hip"
    "live_mode": ("dataset", "live", bool), "keep_live_time": ("dataset", "keep_alive"),
    "dataset_format": ("dataset", "format"), "dataset_dir": ("dataset", "path"),
    # camera frame                                                             cfg.py:35-38
    "mh": ("camera", "mh"), "mw": ("camera", "mw"), "height": ("camera", "h"), "width": ("camera", "w"),
    # model / render                                                           cfg.py:75-95
    "win_size": ("model", "window_size"), "win_size_bg": ("model", "window_size_bg"),
    "keyframe_buffer_size": ("model", "keyframe_buffer_size"),
    "obj_scale": ("model", "obj_scale"), "bg_scale": ("model", "bg_scale"),
    "hidden_feature_size": ("model", "hidden_feature_size"), "hidden_feature_size_bg": ("model", "hidden_feature_size_bg"),
    "clip_point_feature_size": ("model", "clip_point_feature_size"), "n_unidir_funcs": ("model", "n_unidir_funcs"),
    "surface_eps": ("model", "surface_eps"), "stop_eps": ("model", "other_eps"),
    "n_iter_per_frame": ("render", "iters_per_frame"), "n_per_optim": ("render", "n_per_optim"),
    "n_per_optim_bg": ("render", "n_per_optim_bg"), "n_bins_cam2surface": ("render", "n_bins_cam2surface"),
    "n_bins_cam2surface_bg": ("render", "n_bins_cam2surface_bg"), "n_bins": ("render", "n_bins"),
    # visualisation                                                            cfg.py:102-114
    **{name: ("vis", name, bool) for name in ("if_vis", "if_ckpt", "if_render", "if_obj", "save_pcd", "save_mesh")},
    **{name: ("vis", name) for name in ("vis_device", "bg_id", "n_vis_iter", "eps_fine_vis", "n_bins_fine_vis",
                                         "live_voxel_size", "grid_dim")},
}
_DISTORTION_KEYS = ("k1", "k2", "p1", "p2", "k3", "k4", "k5", "k6")


class Config:
    def __init__(self, config_file):
        if isinstance(config_file, dict):
            config = config_file
        else:
            with open(config_file) as json_file:
                config = json.load(json_file)
        for attr, (section, key, *conv) in _PLAIN.items():
            value = config[section][key]
            setattr(self, attr, conv[0](value) if conv else value)
        self.obj_id = -1
        self.depth_scale = 1 / config["trainer"]["scale"]
        self.min_depth, self.max_depth = config["render"]["depth_range"][0], config["render"]["depth_range"][1]
        opt = config["optimizer"]["args"]                                   # cfg.py:98-99
        self.learning_rate, self.weight_decay = opt["lr"], opt["weight_decay"]

        # camera (cfg.py:33-66): the image is cropped by the margins; intrinsics from the JSON or, ScanNet, from the dataset
        cam = config["camera"]
        self.H, self.W = self.height - 2 * self.mh, self.width - 2 * self.mw
        if "fx" in cam:
            fx, fy, cx, cy = cam["fx"], cam["fy"], cam["cx"], cam["cy"]
        else:
            K = load_matrix_from_txt(os.path.join(self.dataset_dir, "intrinsic/intrinsic_depth.txt"))
            fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx - self.mw, cy - self.mh
        if "distortion" in cam:
            self.distortion_array = np.array(cam["distortion"])
        elif "k1" in cam:
            self.distortion_array = np.array([cam[k] for k in _DISTORTION_KEYS])
        else:
            self.distortion_array = None

        # part-level understanding (cfg.py:69-72): part_down exists only when the key is present
        self.part_mode = False
        if "part_mode" in config["trainer"]:
            self.part_mode = bool(config["trainer"]["part_mode"])
            self.part_down = config["trainer"]["part_down"]

        # derived (cfg.py:78-86)
        self.n_samples_per_frame = self.n_per_optim // self.win_size
        self.n_samples_per_frame_bg = self.n_per_optim_bg // self.win_size_bg
        self.keyframe_step = config["model"]["keyframe_step"] / self.stride
        self.keyframe_step_bg = config["model"]["keyframe_step_bg"] / self.stride


def replica_room0_config(train_device="cuda:0", strategy="hip", **overrides):
    """The hyper-parameters of configs/Replica/room_0.json (the shipped Replica configs differ only in
    the dataset path), as a dict accepted by Config.  `overrides` = {"section.key": value}."""
    cfg = {
        "dataset": {"live": 0, "path": "", "format": "Replica", "keep_alive": 20},
        "optimizer": {"args": {"lr": 0.001, "weight_decay": 0.013, "pose_lr": 0.001}},
        "trainer": {"part_mode": 1, "part_down": 5, "imap_mode": 0, "start": 0, "stride": 10, "do_bg": 1,
                    "n_models": 100, "train_device": train_device, "data_device": train_device,
                    "training_strategy": strategy, "epochs": 1000000, "scale": 1000.0},
        "render": {"depth_range": [0.0, 8.0], "n_bins": 9, "n_bins_cam2surface": 1, "n_bins_cam2surface_bg": 5,
                   "iters_per_frame": 100, "n_per_optim": 120, "n_per_optim_bg": 1200},
        "model": {"n_unidir_funcs": 5, "obj_scale": 2.0, "bg_scale": 5.0, "color_scaling": 5.0,
                  "opacity_scaling": 10.0, "gt_scene": 1, "surface_eps": 0.1, "other_eps": 0.05,
                  "keyframe_buffer_size": 20, "keyframe_step": 25, "keyframe_step_bg": 50, "window_size": 5,
                  "window_size_bg": 10, "hidden_layers_block": 1, "hidden_feature_size": 32,
                  "hidden_feature_size_bg": 128, "clip_point_feature_size": 512},
        "camera": {"w": 1200, "h": 680, "fx": 600.0, "fy": 600.0, "cx": 599.5, "cy": 339.5, "mw": 0, "mh": 0},
        "vis": {"if_vis": 0, "if_ckpt": 1, "if_render": 0, "if_obj": 0, "save_pcd": 0, "save_mesh": 1,
                "vis_device": train_device, "bg_id": [0, 2, 3], "n_vis_iter": 9999, "eps_fine_vis": 0.1,
                "n_bins_fine_vis": 10, "im_vis_reduce": 10, "grid_dim": 128, "live_vis": 1,
                "live_voxel_size": 0.005},
    }
    for k, val in overrides.items():
        sec, key = k.split(".")
        cfg[sec][key] = val
    return cfg
