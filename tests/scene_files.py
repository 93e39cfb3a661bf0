"""Test helper: write a tiny scene in the reference's on-disk Replica / ScanNet layouts (dataset.py:43-442):
a textured wall (instance id 1 = background) at 3 m with two boxes in front of it, seen from a camera that pans."""
import os
import pickle

import numpy as np
from PIL import Image

W, H = 64, 48
FX = FY = 60.0
CX, CY = 31.5, 23.5


def _frame(i, rs):
    """-> rgb u8 [H,W,3], depth_mm u16 [H,W], inst u16 [H,W] for camera x-offset 0.02*i m (pure translation)."""
    rgb = np.zeros((H, W, 3), np.uint8)
    depth = np.full((H, W), 3000, np.uint16)
    inst = np.ones((H, W), np.uint16)                      # 1 = wall (background class)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb[..., 0] = 90 + 40 * ((xx // 8 + yy // 8) % 2)
    rgb[..., 1] = 80
    rgb[..., 2] = 70
    sh = i                                                  # boxes drift by one pixel per frame
    a = (slice(10, 34), slice(8 + sh, 28 + sh))             # object 4: 24 x 20 px at 1.5 m
    rgb[a] = (200, 40, 40)
    depth[a] = 1500
    inst[a] = 4
    b = (slice(14, 40), slice(36 + sh, 54 + sh))            # object 7: 26 x 18 px at 2.0 m
    rgb[b] = (40, 60, 210)
    depth[b] = 2000
    inst[b] = 7
    inst[0:3, :] = 0                                        # a strip nobody labelled (-> unknown)
    inst[44:47, 2:6] = 5                                    # a 3 x 4 px speck: too small, must become unknown
    return rgb, depth, inst


def write_scene(root, fmt="Replica", n_frames=40, part_dim=0, part_down=5, seed=0, color_scale=2):
    """Frames 0, 10, 20, ... are the ones a stride-10 loader reads; instance / class maps exist per 10 frames."""
    rs = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    for d in ("depth", "instance_our", "class_our", "rgb" if fmt == "Replica" else "color"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    if part_dim:
        os.makedirs(os.path.join(root, "partlevel"), exist_ok=True)
    traj = []
    clip_all, cap_all = [], []
    for i in range(n_frames):
        rgb, depth, inst = _frame(i // 10, rs)
        if fmt == "Replica":
            Image.fromarray(rgb).save(os.path.join(root, "rgb", "rgb_%d.png" % i))
            Image.fromarray(depth).save(os.path.join(root, "depth", "depth_%d.png" % i))
        else:
            big = np.repeat(np.repeat(rgb, color_scale, axis=0), color_scale, axis=1)   # colour at (twice) the depth resolution
            Image.fromarray(big).save(os.path.join(root, "color", "%d.jpg" % i), quality=95)
            Image.fromarray(depth).save(os.path.join(root, "depth", "%d.png" % i))
        T = np.eye(4)
        T[0, 3] = 0.02 * (i // 10)
        traj.append(T.reshape(-1))
        if i % 10 == 0:
            j = i // 10
            name = "semantic_%s_%d.png" if fmt == "Replica" else "%s%d.png"
            Image.fromarray(inst).save(os.path.join(root, "instance_our", name % ("instance", j) if fmt == "Replica" else "%d.png" % j))
            Image.fromarray(inst).save(os.path.join(root, "class_our", name % ("class", j) if fmt == "Replica" else "%d.png" % j))
            f = lambda k, n: (np.eye(n)[k % n] + 0.01 * rs.randn(n)).astype(np.float32)
            clip_all.append({k: f(k, 16)[None] for k in (1, 4, 7, 5)})       # [1, C] like the CLIP entries
            cap_all.append({k: f(k + 1, 12) for k in (1, 4, 7, 5)})
        if part_dim and i % 10 == 0:
            pf = rs.randn(H // part_down, W // part_down, part_dim).astype(np.float32)
            np.save(os.path.join(root, "partlevel", "%d.npy" % i), pf)
    np.savetxt(os.path.join(root, "traj_w_c.txt"), np.stack(traj), delimiter=" ")
    with open(os.path.join(root, "object_clipfeat.pkl"), "wb") as fh:
        pickle.dump(clip_all, fh)
    with open(os.path.join(root, "object_capfeat.pkl"), "wb") as fh:
        pickle.dump(cap_all, fh)
    return dict(W=W, H=H, fx=FX, fy=FY, cx=CX, cy=CY)


def write_grid_scene(root, fmt="ScanNet", W=640, H=480, n_obj=15, n_frames=20, part_dim=0, part_down=10, seed=0,
                     stored_down=None):
    """A larger scene in the same on-disk layouts: n_obj rectangles (instance ids 2 .. n_obj + 1) on a 5-column grid in
    front of a wall (id 1), each at its own depth -- BASELINE configs[3]'s per-GPU shape (a 640 x 480 ScanNet camera,
    15 objects, part features).  The camera does not move.  Returns the intrinsics.
    stored_down: resolution divisor of the part maps on disk (default part_down; the reference's ScanNet data stores
    them at 1/5 and halves them on load when part_down is 10, dataset.py:305-309)."""
    stored_down = stored_down or part_down
    rs = np.random.RandomState(seed)
    os.makedirs(root, exist_ok=True)
    for d in ("depth", "instance_our", "class_our", "rgb" if fmt == "Replica" else "color"):
        os.makedirs(os.path.join(root, d), exist_ok=True)
    if part_dim:
        os.makedirs(os.path.join(root, "partlevel"), exist_ok=True)
    cols = 5
    rows = (n_obj + cols - 1) // cols
    cw, ch = W // cols, H // rows
    rgb = np.zeros((H, W, 3), np.uint8)
    depth = np.full((H, W), 3500, np.uint16)
    inst = np.ones((H, W), np.uint16)
    yy, xx = np.mgrid[0:H, 0:W]
    rgb[..., 0] = 70 + 30 * ((xx // 16 + yy // 16) % 2)
    rgb[..., 1] = 75
    rgb[..., 2] = 80
    ids = []
    for k in range(n_obj):
        r, c = divmod(k, cols)
        x0, y0 = c * cw + cw // 6, r * ch + ch // 6
        box = (slice(y0, y0 + 2 * ch // 3), slice(x0, x0 + 2 * cw // 3))
        rgb[box] = (40 + 13 * k % 200, 250 - 14 * k, 60 + (37 * k) % 180)
        depth[box] = 1200 + 100 * k
        inst[box] = k + 2
        ids.append(k + 2)
    traj, clip_all, cap_all = [], [], []
    for i in range(n_frames):
        if fmt == "Replica":
            Image.fromarray(rgb).save(os.path.join(root, "rgb", "rgb_%d.png" % i))
            Image.fromarray(depth).save(os.path.join(root, "depth", "depth_%d.png" % i))
        else:
            Image.fromarray(rgb).save(os.path.join(root, "color", "%d.jpg" % i), quality=95)
            Image.fromarray(depth).save(os.path.join(root, "depth", "%d.png" % i))
        traj.append(np.eye(4).reshape(-1))
        if i % 10 == 0:
            j = i // 10
            name = ("semantic_%s_" + str(j) + ".png") if fmt == "Replica" else None
            for sub, tag in (("instance_our", "instance"), ("class_our", "class")):
                Image.fromarray(inst).save(os.path.join(root, sub, (name % tag) if name else "%d.png" % j))
            f = lambda k_, n: (np.eye(n)[k_ % n] + 0.01 * rs.randn(n)).astype(np.float32)
            clip_all.append({k_: f(k_, 16)[None] for k_ in [1] + ids})
            cap_all.append({k_: f(k_ + 1, 12) for k_ in [1] + ids})
            if part_dim:
                pf = rs.randn(H // stored_down, W // stored_down, part_dim).astype(np.float32)
                np.save(os.path.join(root, "partlevel", "%d.npy" % i), pf)
    np.savetxt(os.path.join(root, "traj_w_c.txt"), np.stack(traj), delimiter=" ")
    with open(os.path.join(root, "object_clipfeat.pkl"), "wb") as fh:
        pickle.dump(clip_all, fh)
    with open(os.path.join(root, "object_capfeat.pkl"), "wb") as fh:
        pickle.dump(cap_all, fh)
    return dict(W=W, H=H, fx=0.9 * W, fy=0.9 * W, cx=W / 2 - 0.5, cy=H / 2 - 0.5, ids=ids)
