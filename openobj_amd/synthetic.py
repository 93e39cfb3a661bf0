"""Seeded synthetic Replica-shaped inputs for the object-NeRF path (host side, numpy).

There are no datasets on the build or GPU boxes, so parity tests, the PSNR scene and bench.py
all draw their rays from here.  Everything is generated from `numpy.random.RandomState(seed)` so
the same call yields the same bytes on every machine (torch's CPU/GPU generators are not used).

Two generators:

* `random_batch`   -- shape-faithful random batches (depth U(0.5,6) m with 5 % invalid, labels
                      55/35/10 % = this/other/unknown, stratified + near-surface z-values laid out
                      as `sceneObject.sample_3d_points` does, reference vmap.py:456-554).  Used by
                      the throughput bench and the parity tests.
* `EllipsoidScene` -- K analytic ellipsoids with constant colour and a constant unit feature;
                      rays are intersected analytically, giving ground-truth depth / rgb / labels
                      and held-out rays for PSNR.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict

import numpy as np


def _stratified(rs: np.random.RandomState, lo: np.ndarray, hi: np.ndarray, n_bins: int) -> np.ndarray:
    """lo + range*i/n + U(0,1)*range/n  (reference utils.py:342-379)."""
    lim = np.linspace(0.0, 1.0, n_bins + 1, dtype=np.float32)[:-1]
    rng = (hi - lo).astype(np.float32)
    u = rs.rand(lo.shape[0], n_bins).astype(np.float32)
    return (rng[:, None] * lim[None, :] + lo[:, None] + u * (rng / n_bins)[:, None]).astype(np.float32)


def sample_z(rs: np.random.RandomState, depth: np.ndarray, labels: np.ndarray, n_cam2surf: int,
             n_bins: int, eps: float = 0.1, stop_eps: float = 0.05) -> np.ndarray:
    """Depth-guided z-values for flat arrays of rays (reference vmap.py:483-542)."""
    n = depth.shape[0]
    N, M = n_cam2surf, n_bins
    z = np.zeros((n, N + M), np.float32)
    invalid = depth <= 0.0
    if invalid.any():
        cnt = int(invalid.sum())
        z[invalid] = _stratified(rs, np.zeros(cnt, np.float32),
                                 np.full(cnt, depth.max(), np.float32), N + M)
    valid = ~invalid
    if valid.any():
        d = depth[valid]
        z[valid, :N] = _stratified(rs, np.zeros_like(d), d - eps, N)
        obj = valid & (labels == 1)
        if obj.any():
            g = np.sort(rs.randn(int(obj.sum()), M).astype(np.float32) * np.float32(eps / 3.0), axis=1)
            z[obj, N:] = depth[obj][:, None] + np.clip(g, -eps, eps)
        oth = valid & (labels != 1)
        if oth.any():
            d2 = depth[oth]
            z[oth, N:] = _stratified(rs, d2 - eps, d2 + stop_eps, M)
    return z


def random_batch(K: int, R: int, n_cam2surf: int, n_bins: int, seed: int, feat_dim: int = 0,
                 room: float = 3.0) -> Dict[str, np.ndarray]:
    """Random but shape- and distribution-faithful (K,R,S) training batch.

    Returns float32 arrays: origins [K,R,3], dirs [K,R,3], z [K,R,S], pts [K,R,S,3]
    (= origins + dirs*z, two roundings as reference vmap.py:548), gt_depth [K,R],
    gt_rgb [K,R,3] in [0,1] (uint8/255 as train.py:373), labels [K,R] uint8 and, when
    feat_dim>0, unit-norm gt_feat [K,R,feat_dim].
    """
    rs = np.random.RandomState(seed)
    n = K * R
    depth = rs.uniform(0.5, 6.0, n).astype(np.float32)
    depth[rs.rand(n) < 0.05] = 0.0
    lab = rs.rand(n)
    labels = np.where(lab < 0.55, 1, np.where(lab < 0.90, 0, 2)).astype(np.uint8)
    z = sample_z(rs, depth, labels, n_cam2surf, n_bins)
    origins = rs.uniform(-room, room, (n, 3)).astype(np.float32)
    dirs = np.ones((n, 3), np.float32)
    dirs[:, 0] = rs.uniform(-1.0, 1.0, n)
    dirs[:, 1] = rs.uniform(-0.57, 0.57, n)
    rot = _random_rotations(rs, n)
    dirs = np.einsum("nij,nj->ni", rot, dirs).astype(np.float32)
    # keep the sample points inside the +-4 m box the reference scenes live in
    far = z.max(axis=1)
    tip = origins + dirs * far[:, None]
    shrink = np.maximum(1.0, np.abs(tip).max(axis=1) / 4.0).astype(np.float32)
    dirs = (dirs / shrink[:, None]).astype(np.float32)
    pts = (origins[:, None, :] + (dirs[:, None, :] * z[:, :, None]).astype(np.float32)).astype(np.float32)
    rgb = (rs.randint(0, 256, (n, 3)).astype(np.float32) / np.float32(255.0)).astype(np.float32)
    S = n_cam2surf + n_bins
    out = dict(origins=origins.reshape(K, R, 3), dirs=dirs.reshape(K, R, 3), z=z.reshape(K, R, S),
               pts=pts.reshape(K, R, S, 3), gt_depth=depth.reshape(K, R), gt_rgb=rgb.reshape(K, R, 3),
               labels=labels.reshape(K, R))
    if feat_dim:
        f = rs.randn(n, feat_dim).astype(np.float32)
        f /= np.linalg.norm(f, axis=1, keepdims=True)
        out["gt_feat"] = f.reshape(K, R, feat_dim).astype(np.float32)
    return out


def _random_rotations(rs: np.random.RandomState, n: int) -> np.ndarray:
    q = rs.randn(n, 4)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    return np.stack([
        np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)], -1),
        np.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)], -1),
        np.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)], 1)


@dataclass
class EllipsoidScene:
    """K analytic ellipsoids, one per object network, each centred at the object origin."""
    radii: np.ndarray      # [K,3]
    color: np.ndarray      # [K,3] in [0,1], quantised to uint8/255
    feat: np.ndarray       # [K,C] unit vectors
    bg_color: np.ndarray   # [3]

    @staticmethod
    def make(K: int, feat_dim: int = 512, seed: int = 7) -> "EllipsoidScene":
        rs = np.random.RandomState(seed)
        radii = rs.uniform(0.25, 0.6, (K, 3)).astype(np.float32)
        color = (rs.randint(30, 226, (K, 3)).astype(np.float32) / np.float32(255.0))
        feat = rs.randn(K, feat_dim).astype(np.float32)
        feat /= np.linalg.norm(feat, axis=1, keepdims=True)
        return EllipsoidScene(radii, color.astype(np.float32), feat.astype(np.float32),
                              np.array([0.1, 0.1, 0.1], np.float32))

    @property
    def K(self) -> int:
        return self.radii.shape[0]

    def rays(self, R: int, seed: int):
        """R rays per object: camera on a shell of radius 1.5-2.5 m looking roughly at the object,
        un-normalised directions with unit camera-z (reference vmap.py:701-720)."""
        rs = np.random.RandomState(seed)
        K = self.K
        n = K * R
        o = rs.randn(n, 3)
        o /= np.linalg.norm(o, axis=1, keepdims=True)
        o *= rs.uniform(1.5, 2.5, (n, 1))
        look = -o / np.linalg.norm(o, axis=1, keepdims=True)
        tmp = np.where(np.abs(look[:, 2:3]) < 0.9, np.array([[0, 0, 1.0]]), np.array([[1.0, 0, 0]]))
        right = np.cross(look, tmp)
        right /= np.linalg.norm(right, axis=1, keepdims=True)
        up = np.cross(right, look)
        uv = rs.uniform(-0.28, 0.28, (n, 2))
        d = look + uv[:, :1] * right + uv[:, 1:] * up
        return o.astype(np.float32).reshape(K, R, 3), d.astype(np.float32).reshape(K, R, 3)

    def intersect(self, o: np.ndarray, d: np.ndarray):
        """First hit parameter t (o + t d on the ellipsoid), or 0 where the ray misses."""
        r = self.radii[:, None, :].astype(np.float64)
        oo = o.astype(np.float64) / r
        dd = d.astype(np.float64) / r
        a = (dd * dd).sum(-1)
        b = 2 * (oo * dd).sum(-1)
        c = (oo * oo).sum(-1) - 1.0
        disc = b * b - 4 * a * c
        hit = disc > 0
        t = np.where(hit, (-b - np.sqrt(np.maximum(disc, 0))) / (2 * a), 0.0)
        hit &= t > 0
        return np.where(hit, t, 0.0).astype(np.float32), hit

    def batch(self, R: int, n_cam2surf: int, n_bins: int, seed: int, with_feat: bool = False,
              unknown_frac: float = 0.05) -> Dict[str, np.ndarray]:
        """Training batch (K,R,S).  Hit rays: label 1, depth = hit t, colour = object colour.
        Miss rays: label 0, depth = a far wall at t=3.5, colour = background.  A few rays are
        marked unknown (2)."""
        o, d = self.rays(R, seed)
        K = self.K
        t, hit = self.intersect(o, d)
        rs = np.random.RandomState(seed + 100003)
        depth = np.where(hit, t, np.float32(3.5)).astype(np.float32)
        labels = np.where(hit, 1, 0).astype(np.uint8)
        labels[rs.rand(K, R) < unknown_frac] = 2
        z = sample_z(rs, depth.reshape(-1), labels.reshape(-1), n_cam2surf, n_bins)
        S = n_cam2surf + n_bins
        z = z.reshape(K, R, S)
        pts = (o[:, :, None, :] + (d[:, :, None, :] * z[..., None]).astype(np.float32)).astype(np.float32)
        rgb = np.where(hit[..., None], self.color[:, None, :], self.bg_color[None, None, :]).astype(np.float32)
        out = dict(origins=o, dirs=d, z=z, pts=pts, gt_depth=depth, gt_rgb=rgb, labels=labels, hit=hit)
        if with_feat:
            out["gt_feat"] = np.broadcast_to(self.feat[:, None, :], (K, R, self.feat.shape[1])).copy()
        return out

    def eval_rays(self, R: int, n_samples: int, seed: int = 999) -> Dict[str, np.ndarray]:
        """Held-out rays that hit the object, with `n_samples` mid-point z-values on
        [t-0.3, t+0.3] for rendering; ground truth colour = object colour."""
        o, d = self.rays(R * 8, seed)
        t, hit = self.intersect(o, d)
        K = self.K
        oo = np.zeros((K, R, 3), np.float32)
        dd = np.zeros((K, R, 3), np.float32)
        tt = np.zeros((K, R), np.float32)
        for k in range(K):
            idx = np.nonzero(hit[k])[0][:R]
            assert idx.size == R, "not enough hitting rays"
            oo[k], dd[k], tt[k] = o[k, idx], d[k, idx], t[k, idx]
        edges = np.linspace(-0.3, 0.3, n_samples + 1, dtype=np.float32)
        mid = 0.5 * (edges[1:] + edges[:-1])
        z = (tt[..., None] + mid[None, None, :]).astype(np.float32)
        pts = (oo[:, :, None, :] + (dd[:, :, None, :] * z[..., None]).astype(np.float32)).astype(np.float32)
        rgb = np.broadcast_to(self.color[:, None, :], (K, R, 3)).astype(np.float32).copy()
        return dict(origins=oo, dirs=dd, z=z, pts=pts, gt_depth=tt, gt_rgb=rgb)


def grid_frame(i: int, n_objects: int, W: int = 1200, H: int = 680, part: bool = False) -> Dict[str, object]:
    """Frame i of a synthetic stream for the mapping loop (openobj_amd/mapping.py) at the reference's camera size:
    n_objects rectangles on a grid in front of a wall, instance image / depth / colour / 2-D boxes laid out as the
    dataset adapters hand them over ([W, H] arrays, dataset.py).  tools/mapping_bench.py and bench.py's `native_frame`."""
    import torch
    nx = int(np.ceil(np.sqrt(n_objects * W / H)))
    ny = int(np.ceil(n_objects / nx))
    cw, ch = W // nx, H // ny
    inst = np.zeros((W, H), np.int32)               # 0 = background
    depth = np.full((W, H), 3.0, np.float32)
    rgb = np.full((W, H, 3), 90, np.uint8)
    bbox = {0: torch.tensor([0, W, 0, H])}
    k = 0
    for iy in range(ny):
        for ix in range(nx):
            if k >= n_objects:
                break
            x0, y0 = ix * cw + cw // 6 + i, iy * ch + ch // 6
            x1, y1 = x0 + 2 * cw // 3, y0 + 2 * ch // 3
            oid = k + 4
            inst[x0:x1, y0:y1] = oid
            depth[x0:x1, y0:y1] = 1.2 + 0.02 * k
            rgb[x0:x1, y0:y1] = ((37 * k) % 255, (91 * k) % 255, (53 * k) % 255)
            bbox[oid] = torch.tensor([max(x0 - 8, 0), min(x1 + 8, W - 1), max(y0 - 8, 0), min(y1 + 8, H - 1)])
            k += 1
    T = np.eye(4)
    T[0, 3] = 0.002 * i
    feats = {oid: np.ones((1, 8), np.float32) for oid in bbox}
    s = {"image": rgb, "depth": depth, "T": T, "obj": inst, "bbox_dict": bbox, "frame_id": 10 * i,
         "obj_clip": feats, "obj_cap": {o: np.ones(8, np.float32) for o in bbox}}
    if part:
        s["part_feat"] = torch.randn(W // 5, H // 5, 512)
    return s
