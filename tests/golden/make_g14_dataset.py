#!/usr/bin/env python3
"""Fixture G14: the reference's OWN dataset classes (dataset.py:43-442, Replica and ScanNet)
run on the helper scene of tests/scene_files.py; every field of every sample they return is stored.

Run in the build container only (needs /root/reference):   python tests/golden/make_g14_dataset.py

dataset.py imports packages this image does not have.  They are replaced, for this script only, by stand-ins:
  cv2          imread = Pillow decode (colour returned in BGR order like OpenCV, 16-bit PNGs unchanged), cvtColor =
               channel flip, resize = identity (the G14 ScanNet scene stores colour at the depth resolution, so the
               reference's resize call does not resample; OpenCV's fixed-point INTER_LINEAR cannot be reproduced
               here and stays a documented restatement, openobj_amd/dataset.py resize_linear)
  torchvision  transforms.Compose = function composition
  natsort      natsorted = natural ordering of the digit runs
  imgviz, open3d   unused by the sample path: MagicMock
So G14 pins the reference's LOGIC on decoded pixels: file naming and indexing, transposition, depth scaling / filtering,
the instance -> object map rules, box extraction and enlargement, feature dictionaries, part features -- not the
image decoders.
"""
import os
import re
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/objnerf"
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)

cv2 = types.ModuleType("cv2")
cv2.IMREAD_UNCHANGED, cv2.COLOR_BGR2RGB, cv2.INTER_LINEAR, cv2.INTER_NEAREST = -1, 4, 1, 0


def _imread(path, flags=1):
    with Image.open(path) as im:
        if flags == -1:
            a = np.asarray(im)
            return a[..., ::-1].copy() if a.ndim == 3 else a.copy()
        return np.asarray(im.convert("RGB"))[..., ::-1].copy()


def _resize(img, dsize, interpolation=1):
    assert tuple(dsize) == (img.shape[1], img.shape[0]), "G14 scenes keep colour at the depth resolution"
    return img.copy()


cv2.imread = _imread
cv2.cvtColor = lambda img, code: img[..., ::-1].copy()
cv2.resize = _resize
sys.modules["cv2"] = cv2
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")


class Compose:
    def __init__(self, fs):
        self.fs = fs

    def __call__(self, x):
        for f in self.fs:
            x = f(x)
        return x


tvt.Compose = Compose
tv.transforms = tvt
sys.modules["torchvision"], sys.modules["torchvision.transforms"] = tv, tvt
ns = types.ModuleType("natsort")
ns.natsorted = lambda xs: sorted(xs, key=lambda p: [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", p)])
sys.modules["natsort"] = ns
for name in ["imgviz", "open3d", "trimesh", "bidict", "skimage", "skimage.measure", "matplotlib", "matplotlib.pyplot"]:
    sys.modules.setdefault(name, MagicMock())

import dataset as ref_dataset          # noqa: E402  (the reference)
import scene_files as SF               # noqa: E402

CASES = [("Replica", dict(n_frames=30, part_dim=4, part_down=4)),
         ("ScanNet", dict(n_frames=30, part_dim=4, part_down=4, color_scale=1)),
         ("ScanNet", dict(n_frames=20, part_dim=6, part_down=2, color_scale=1, cfg_part_down=10))]


def ref_cfg(root, fmt, part_down):
    return types.SimpleNamespace(dataset_format=fmt, imap_mode=0, start=0, stride=10, dataset_dir=root,
                                 depth_scale=1 / 1000.0, max_depth=8.0, part_mode=1, part_down=part_down,
                                 W=SF.W, H=SF.H, fx=SF.FX, fy=SF.FY, cx=SF.CX, cy=SF.CY)


def main():
    out = {}
    for ci, (fmt, kw) in enumerate(CASES):
        kw = dict(kw)
        cfg_pd = kw.pop("cfg_part_down", kw["part_down"])
        with tempfile.TemporaryDirectory() as root:
            SF.write_scene(root, fmt, seed=3 + ci, **kw)
            ds = (ref_dataset.Replica if fmt == "Replica" else ref_dataset.ScanNet)(ref_cfg(root, fmt, cfg_pd))
            n = len(ds)
            out[f"c{ci}_len"] = n
            for i in range(n):
                s = ds[i]
                pre = f"c{ci}_s{i}_"
                out[pre + "image"] = np.asarray(s["image"])
                out[pre + "depth"] = np.asarray(s["depth"])
                out[pre + "T"] = np.asarray(s["T"])
                out[pre + "T_obj"] = np.asarray(s["T_obj"])
                out[pre + "obj"] = np.asarray(s["obj"])
                out[pre + "frame_id"] = np.asarray(s["frame_id"])
                keys = sorted(int(k) for k in s["bbox_dict"])
                out[pre + "keys"] = np.asarray(keys, np.int64)
                out[pre + "boxes"] = np.stack([np.asarray(s["bbox_dict"][k]) for k in s["bbox_dict"]
                                               ])[np.argsort([int(k) for k in s["bbox_dict"]])]
                out[pre + "clip"] = np.stack([np.asarray(s["obj_clip"][k]).reshape(-1) for k in keys])
                out[pre + "cap"] = np.stack([np.asarray(s["obj_cap"][k]).reshape(-1) for k in keys])
                assert sorted(int(k) for k in s["obj_clip"]) == keys == sorted(int(k) for k in s["obj_cap"])
                out[pre + "part_feat"] = s["part_feat"].numpy()
    path = os.path.join(HERE, "g14_dataset.npz")
    np.savez_compressed(path, **out)
    print("g14_dataset: %.1f KiB, %d arrays" % (os.path.getsize(path) / 1024, len(out)))


if __name__ == "__main__":
    main()
