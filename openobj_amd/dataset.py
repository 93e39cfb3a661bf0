"""Dataset adapters for the on-disk layouts the reference trains from:
`Replica` and `ScanNet` of the reference's dataset.py:43-442, as written by its mask_graph.py / sam_clip_dir.py
preprocessing.  Host-side file I/O only -- nothing here touches the GPU.

Every sample is the reference's dict (dataset.py:175-190, 404-412), arrays TRANSPOSED to [W, H] like there:
    image u8 [W,H,3] (RGB) | depth f32 [W,H] metres, > max_depth zeroed | T f64 [4,4] camera->world | T_obj eye(4)
    obj i32 [W,H]: instance id per pixel, 0 = background, -1 = unknown
    bbox_dict {id: int64 tensor [w_min, w_max, h_min, h_max]} (object boxes enlarged by 20 %) | frame_id
    obj_clip / obj_cap {id: feature}      (+ part_feat [W',H',C] in part mode)

Images are decoded with Pillow (the reference uses OpenCV, which this build does not depend on); PNG / JPEG decode
to the same pixels.  The one resampling step -- ScanNet colour frames resized to the depth resolution
(dataset.py:313) -- is OpenCV's INTER_LINEAR rule (half-pixel centres, no antialiasing) written in numpy; OpenCV
evaluates it in 11-bit fixed point, so single values may differ by one grey level.
"""
import glob
import math
import os
import pickle
import re

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

BBOX_SCALE = 0.2          # dataset.py:70,255
MIN_BOX_SIDE = 10         # dataset.py:147


def _read_image(path):
    from PIL import Image          # Pillow; imported lazily so the module loads without it
    with Image.open(path) as im:
        if im.mode in ("I;16", "I;16B", "I;16L", "I"):
            return np.asarray(im).astype(np.int32)
        if im.mode in ("L", "P"):
            return np.asarray(im)
        return np.asarray(im.convert("RGB"))


def _natural_key(path):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", os.path.basename(path))]


def resize_linear(img, out_w, out_h):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for a u8 [H, W, C] image."""
    in_h, in_w = img.shape[:2]
    if (in_h, in_w) == (out_h, out_w):
        return img

    def taps(n_in, n_out):
        x = (np.arange(n_out, dtype=np.float64) + 0.5) * (n_in / n_out) - 0.5
        x0 = np.floor(x).astype(np.int64)
        f = x - x0
        lo = np.clip(x0, 0, n_in - 1)
        hi = np.clip(x0 + 1, 0, n_in - 1)
        return lo, hi, f

    ylo, yhi, fy = taps(in_h, out_h)
    xlo, xhi, fx = taps(in_w, out_w)
    a = img.astype(np.float64)
    top = a[ylo][:, xlo] * (1 - fx)[None, :, None] + a[ylo][:, xhi] * fx[None, :, None]
    bot = a[yhi][:, xlo] * (1 - fx)[None, :, None] + a[yhi][:, xhi] * fx[None, :, None]
    out = top * (1 - fy)[:, None, None] + bot * fy[:, None, None]
    return np.clip(np.rint(out), 0, 255).astype(np.uint8)


def enlarge_bbox(bbox, scale, w, h):
    """utils.py:64-87: grow (min_x, min_y, max_x, max_y) by scale/2 of its size per side, clipped to the image;
    None when a margin rounds to zero."""
    min_x, min_y, max_x, max_y = [int(v) for v in bbox]
    margin_x = int(0.5 * scale * (max_x - min_x))
    margin_y = int(0.5 * scale * (max_y - min_y))
    if margin_x == 0 or margin_y == 0:
        return None
    return [int(np.clip(min_x - margin_x, 0, w - 1)), int(np.clip(min_y - margin_y, 0, h - 1)),
            int(np.clip(max_x + margin_x, 0, w - 1)), int(np.clip(max_y + margin_y, 0, h - 1))]


def mask_extent(mask):
    """utils.py:109-121 for one [A, B] mask: (first, one-past-last) index with a True along each axis."""
    a = np.flatnonzero(mask.any(axis=1))
    b = np.flatnonzero(mask.any(axis=0))
    return int(a[0]), int(a[-1]) + 1, int(b[0]), int(b[-1]) + 1


def frame_objects(inst, cls_of, background_cls, clip_this, cap_this, bbox_scale=BBOX_SCALE):
    """The per-frame object bookkeeping shared by both formats (dataset.py:112-173, 337-399).

    inst: i32 [W, H] instance ids with 0 already mapped to -1 (unknown).  cls_of(mask) -> the semantic classes
    under a mask.  Returns (obj map, bbox_dict, clip_dict, cap_dict); obj = instance ids with the background
    classes set to 0 and every pixel that is not in a kept object mask set to -1."""
    W, H = inst.shape
    kept = np.zeros_like(inst)
    bbox_dict, clip_dict, cap_dict = {}, {}, {}
    for inst_id in np.unique(inst):
        if inst_id == -1:
            continue
        m = inst == inst_id
        sem = np.unique(cls_of(m))
        assert sem.shape[0] != 0
        if any(int(s) in background_cls for s in sem):
            continue
        w0, w1, h0, h1 = mask_extent(m)                 # axis 0 is image x (arrays are [W, H])
        if (w1 - w0) <= MIN_BOX_SIDE or (h1 - h0) <= MIN_BOX_SIDE:
            continue                                    # too small: stays unknown
        # (the reference passes w = shape[1], h = shape[0] for its [W, H] arrays: dataset.py:152-153)
        big = enlarge_bbox([h0, w0, h1, w1], scale=bbox_scale, w=H, h=W)
        if big is None:
            continue
        kept[m] = 1
        key = int(inst_id)
        bbox_dict[key] = torch.from_numpy(np.array([big[1], big[3], big[0], big[2]], dtype=np.int64))
        clip_dict[key] = clip_this[key]
        cap_dict[key] = cap_this[key]
    obj = inst.copy()
    for s in background_cls:
        obj[obj == s] = 0
    obj[(kept == 0) & (obj != 0)] = -1
    return obj, bbox_dict, clip_dict, cap_dict


class _FrameSet(Dataset):
    background_cls_list = [1]          # "wall" in the preprocessed labels (dataset.py:68,254)

    def _common_init(self, cfg):
        self.imap_mode = cfg.imap_mode
        self.start = cfg.start
        self.stride = cfg.stride
        self.root_dir = cfg.dataset_dir
        self.Twc = np.loadtxt(os.path.join(self.root_dir, "traj_w_c.txt"), delimiter=" ").reshape([-1, 4, 4])
        with open(os.path.join(self.root_dir, "object_clipfeat.pkl"), "rb") as f:
            self.obj_clipfeat = pickle.load(f)
        with open(os.path.join(self.root_dir, "object_capfeat.pkl"), "rb") as f:
            self.obj_capfeat = pickle.load(f)
        self.depth_scale = cfg.depth_scale
        self.max_depth = cfg.max_depth
        self.part_mode = cfg.part_mode
        self.part_down = getattr(cfg, "part_down", 1)
        self.bbox_scale = BBOX_SCALE

    def _depth(self, raw):
        d = raw.astype(np.float32) * np.float32(self.depth_scale)      # image_transforms.DepthScale
        d[d > self.max_depth] = 0.0                                     # image_transforms.DepthFilter
        return d

    def _finish(self, image, depth, inst, sem, idx, idx_no, uses_class_map, part_feat):
        clip_this, cap_this = self.obj_clipfeat[idx_no], self.obj_capfeat[idx_no]
        bbox_dict, clip_dict, cap_dict = {}, {}, {}
        if self.imap_mode:
            obj = np.zeros(depth.shape, dtype=np.int32)
        else:
            inst = inst.copy()
            inst[inst == 0] = -1
            if uses_class_map:
                sem = sem.copy()
                sem[sem == 0] = -1
                cls_of = lambda m: sem[m]                # Replica: class image under the mask (dataset.py:124)
            else:
                cls_of = lambda m: inst[m]               # ScanNet: the instance id itself (dataset.py:351)
            obj, bbox_dict, clip_dict, cap_dict = frame_objects(inst, cls_of, self.background_cls_list, clip_this,
                                                                cap_this, self.bbox_scale)
        if 1 in clip_this:                               # the background is visible: id 0, whole image
            bbox_dict[0] = torch.from_numpy(np.array([0, int(obj.shape[0]), 0, int(obj.shape[1])], dtype=np.int64))
            clip_dict[0] = clip_this[1]
            cap_dict[0] = cap_this[1]
        sample = {"image": image, "depth": depth, "T": self.Twc[idx], "T_obj": np.eye(4), "obj": obj,
                  "bbox_dict": bbox_dict, "frame_id": idx, "obj_clip": clip_dict, "obj_cap": cap_dict}
        if self.part_mode:
            sample["part_feat"] = part_feat
        return sample

    def _part_feat(self, idx):
        if not self.part_mode:
            return None
        return torch.tensor(np.load(os.path.join(self.root_dir, "partlevel", str(idx) + ".npy")).transpose((1, 0, 2)))


class Replica(_FrameSet):
    """dataset.py:43-190: rgb/rgb_<i>.png, depth/depth_<i>.png (u16), instance_our/semantic_instance_<i//10>.png,
    class_our/semantic_class_<i//10>.png, traj_w_c.txt, object_{clip,cap}feat.pkl, partlevel/<i>.npy."""

    def __init__(self, cfg):
        self._common_init(cfg)

    def __len__(self):
        return int((len(os.listdir(os.path.join(self.root_dir, "depth"))) - self.start) / self.stride)

    def __getitem__(self, i):
        idx = int(self.start + i * self.stride)
        idx_no = int(idx / 10)
        r = self.root_dir
        depth = self._depth(_read_image(os.path.join(r, "depth", "depth_%d.png" % idx)).transpose(1, 0))
        image = _read_image(os.path.join(r, "rgb", "rgb_%d.png" % idx)).astype(np.uint8).transpose(1, 0, 2)
        sem = _read_image(os.path.join(r, "class_our", "semantic_class_%d.png" % idx_no)).astype(np.int32).transpose(1, 0)
        inst = _read_image(os.path.join(r, "instance_our", "semantic_instance_%d.png" % idx_no)).astype(np.int32).transpose(1, 0)
        return self._finish(image, depth, inst, sem, idx, idx_no, True, self._part_feat(idx))


class ScanNet(_FrameSet):
    """dataset.py:192-442: color/<i>.jpg, depth/<i>.png, instance_our/*.png and class_our/*.png (natural order, one
    per 10 frames); colour is resized to the depth resolution; part features optionally halved (part_down 10)."""

    def __init__(self, cfg):
        self._common_init(cfg)
        num = lambda p: int(os.path.basename(p)[:-4])
        self.color_paths = sorted(glob.glob(os.path.join(self.root_dir, "color", "*.jpg")), key=num)
        self.depth_paths = sorted(glob.glob(os.path.join(self.root_dir, "depth", "*.png")), key=num)
        self.inst_paths = sorted(glob.glob(os.path.join(self.root_dir, "instance_our", "*.png")), key=_natural_key)
        self.sem_paths = sorted(glob.glob(os.path.join(self.root_dir, "class_our", "*.png")), key=_natural_key)
        self.n_img = len(self.color_paths)

    def __len__(self):
        return math.ceil((self.n_img - self.start) / self.stride)

    def _part_feat(self, idx):
        pf = super()._part_feat(idx)
        if pf is not None and self.part_down == 10:      # dataset.py:305-309
            pf = pf.permute(2, 0, 1).unsqueeze(0)
            pf = torch.nn.functional.interpolate(pf, scale_factor=0.5, mode="bilinear", align_corners=False)
            pf = pf.squeeze(0).permute(1, 2, 0)
        return pf

    def __getitem__(self, i):
        idx = int(self.start + i * self.stride)
        idx_no = int(idx / 10)
        color = _read_image(self.color_paths[idx]).astype(np.uint8).transpose(1, 0, 2)      # [W, H, 3]
        raw = np.nan_to_num(_read_image(self.depth_paths[idx]).astype(np.float32).transpose(1, 0), nan=0.0)
        # the reference calls cv2.resize(color [W,H,3], (W_d, H_d)) with (H_d, W_d) = raw.shape, i.e. dsize = raw.shape
        # reversed: the result has raw's shape (dataset.py:312-313)
        color = resize_linear(color, raw.shape[1], raw.shape[0])
        depth = self._depth(raw)
        inst = sem = None
        if not self.imap_mode:
            sem = _read_image(self.sem_paths[idx_no]).astype(np.int32).transpose(1, 0)
            inst = _read_image(self.inst_paths[idx_no]).astype(np.int32).transpose(1, 0)
        return self._finish(color, depth, inst, sem, idx, idx_no, False, self._part_feat(idx))


def init_loader(cfg, multi_worker=True):
    """dataset.py:20-41.  batch_size None: the loader yields one frame dict at a time, arrays as tensors."""
    if cfg.dataset_format == "Replica":
        ds = Replica(cfg)
    elif cfg.dataset_format == "ScanNet":
        ds = ScanNet(cfg)
    else:
        raise ValueError("Dataset format {} not found".format(cfg.dataset_format))
    if multi_worker:
        return DataLoader(ds, batch_size=None, shuffle=False, num_workers=4, pin_memory=True, prefetch_factor=2,
                          persistent_workers=True)
    return DataLoader(ds, batch_size=None, shuffle=False, num_workers=0)
