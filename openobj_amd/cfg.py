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


class Config:
    def __init__(self, config_file):
        if isinstance(config_file, dict):
            config = config_file
        else:
            with open(config_file) as json_file:
                config = json.load(json_file)

        # training strategy                                                    cfg.py:16-26
        self.start = config["trainer"]["start"]
        self.stride = config["trainer"]["stride"]
        self.do_bg = bool(config["trainer"]["do_bg"])
        self.training_device = config["trainer"]["train_device"]
        self.data_device = config["trainer"]["data_device"]
        self.max_n_models = config["trainer"]["n_models"]
        self.live_mode = bool(config["dataset"]["live"])
        self.keep_live_time = config["dataset"]["keep_alive"]
        self.imap_mode = config["trainer"]["imap_mode"]
        self.training_strategy = config["trainer"]["training_strategy"]  # "forloop" "vmap" "hip"
        self.obj_id = -1

        # dataset setting                                                      cfg.py:29-31
        self.dataset_format = config["dataset"]["format"]
        self.dataset_dir = config["dataset"]["path"]
        self.depth_scale = 1 / config["trainer"]["scale"]
        # camera setting                                                       cfg.py:33-66
        self.max_depth = config["render"]["depth_range"][1]
        self.min_depth = config["render"]["depth_range"][0]
        self.mh = config["camera"]["mh"]
        self.mw = config["camera"]["mw"]
        self.height = config["camera"]["h"]
        self.width = config["camera"]["w"]
        self.H = self.height - 2 * self.mh
        self.W = self.width - 2 * self.mw
        if "fx" in config["camera"]:
            self.fx = config["camera"]["fx"]
            self.fy = config["camera"]["fy"]
            self.cx = config["camera"]["cx"] - self.mw
            self.cy = config["camera"]["cy"] - self.mh
        else:   # ScanNet: intrinsics from file (cfg.py:46-51)
            intrinsic = load_matrix_from_txt(os.path.join(self.dataset_dir, "intrinsic/intrinsic_depth.txt"))
            self.fx = intrinsic[0, 0]
            self.fy = intrinsic[1, 1]
            self.cx = intrinsic[0, 2] - self.mw
            self.cy = intrinsic[1, 2] - self.mh
        if "distortion" in config["camera"]:
            self.distortion_array = np.array(config["camera"]["distortion"])
        elif "k1" in config["camera"]:
            c = config["camera"]
            self.distortion_array = np.array([c["k1"], c["k2"], c["p1"], c["p2"], c["k3"], c["k4"], c["k5"], c["k6"]])
        else:
            self.distortion_array = None

        # part-level understanding                                             cfg.py:69-72
        self.part_mode = False
        if "part_mode" in config["trainer"]:
            self.part_mode = bool(config["trainer"]["part_mode"])
            self.part_down = config["trainer"]["part_down"]

        # training setting                                                     cfg.py:75-95
        self.win_size = config["model"]["window_size"]
        self.n_iter_per_frame = config["render"]["iters_per_frame"]
        self.n_per_optim = config["render"]["n_per_optim"]
        self.n_samples_per_frame = self.n_per_optim // self.win_size
        self.win_size_bg = config["model"]["window_size_bg"]
        self.n_per_optim_bg = config["render"]["n_per_optim_bg"]
        self.n_samples_per_frame_bg = self.n_per_optim_bg // self.win_size_bg
        self.keyframe_buffer_size = config["model"]["keyframe_buffer_size"]
        self.keyframe_step = config["model"]["keyframe_step"] / self.stride
        self.keyframe_step_bg = config["model"]["keyframe_step_bg"] / self.stride
        self.obj_scale = config["model"]["obj_scale"]
        self.bg_scale = config["model"]["bg_scale"]
        self.hidden_feature_size = config["model"]["hidden_feature_size"]
        self.hidden_feature_size_bg = config["model"]["hidden_feature_size_bg"]
        self.clip_point_feature_size = config["model"]["clip_point_feature_size"]
        self.n_bins_cam2surface = config["render"]["n_bins_cam2surface"]
        self.n_bins_cam2surface_bg = config["render"]["n_bins_cam2surface_bg"]
        self.n_bins = config["render"]["n_bins"]
        self.n_unidir_funcs = config["model"]["n_unidir_funcs"]
        self.surface_eps = config["model"]["surface_eps"]
        self.stop_eps = config["model"]["other_eps"]

        # optimizer setting                                                    cfg.py:98-99
        self.learning_rate = config["optimizer"]["args"]["lr"]
        self.weight_decay = config["optimizer"]["args"]["weight_decay"]

        # vis setting                                                          cfg.py:102-114
        v = config["vis"]
        self.if_vis = bool(v["if_vis"])
        self.if_ckpt = bool(v["if_ckpt"])
        self.if_render = bool(v["if_render"])
        self.if_obj = bool(v["if_obj"])
        self.save_pcd = bool(v["save_pcd"])
        self.save_mesh = bool(v["save_mesh"])
        self.vis_device = v["vis_device"]
        self.bg_id = v["bg_id"]
        self.n_vis_iter = v["n_vis_iter"]
        self.eps_fine_vis = v["eps_fine_vis"]
        self.n_bins_fine_vis = v["n_bins_fine_vis"]
        self.live_voxel_size = v["live_voxel_size"]
        self.grid_dim = v["grid_dim"]


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
