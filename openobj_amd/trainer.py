"""Trainer -- owns one OccupancyMap + UniDirsEmbed per object (trainer.py:11-128); both modules'
parameters are views of ONE per-object arena block, so update_vmap can gather K of them with a copy."""
import numpy as np
import torch

from . import embedding, model, ops


class Trainer:
    def __init__(self, cfg):
        self.obj_id = cfg.obj_id
        self.device = cfg.training_device
        self.hidden_feature_size = cfg.hidden_feature_size        # 32 objects / 128 background
        self.clip_point_feature_size = cfg.clip_point_feature_size
        self.obj_scale = cfg.obj_scale
        self.n_unidir_funcs = cfg.n_unidir_funcs
        self.emb_size1 = 21 * (3 + 1) + 3                         # trainer.py:20
        self.emb_size2 = 21 * (5 + 1) + 3 - self.emb_size1        # trainer.py:21
        self.load_network()
        self.bound_extent = 0.995 if self.obj_id == 0 else 0.9    # trainer.py:25-28
        self.W_vis, self.H_vis = cfg.W, cfg.H
        self.T_WC_gt = None
        self.dirs_C_gt = None
        self.input_pcs = None

    def load_network(self):                                       # trainer.py:36-44
        self.arena = ops.ParamArena(1, ops.NetShape(self.hidden_feature_size, self.clip_point_feature_size,
                                                    self.n_unidir_funcs + 1), self.device)
        self.fc_occ_map = model.OccupancyMap(self.emb_size1, self.emb_size2, hidden_size=self.hidden_feature_size,
                                             clip_size=self.clip_point_feature_size, device=self.device,
                                             _arena=self.arena)
        self.fc_occ_map.apply(model.init_weights)
        self.pe = embedding.UniDirsEmbed(max_deg=self.n_unidir_funcs, scale=self.obj_scale, device=self.device,
                                         _arena=self.arena)

    def eval_points(self, points, chunk_size=300000):
        """points [N,3] -> (occupancy [N], color [N,3], clip [N,C]) or None (trainer.py:105-128).
        One fused launch per chunk: PE + MLP + feature head; occupancy = sigmoid(alpha)."""
        self.arena.scale.fill_(float(self.obj_scale))
        n_chunks = int(np.ceil(points.shape[0] / chunk_size))
        occ, color, clip = [], [], []
        with torch.no_grad():
            for k in range(n_chunks):
                pts = points[k * chunk_size:(k + 1) * chunk_size].reshape(1, -1, 3).to(self.device).contiguous()
                a, c, _, f = ops.eval_points(self.arena, pts, want_clip=True)
                occ.append(ops.occupancy(a[0]))
                color.append(c[0])
                clip.append(f[0])
        occ, color, clip = torch.cat(occ), torch.cat(color), torch.cat(clip)
        if occ.max() == 0:
            print("no occ")
            return None
        return (occ, color, clip)

    def sample_points_bbox(self, bbox, do_eval=True, draws=None):
        """Points inside a known oriented 3-D box along the rays self.T_WC_gt / self.dirs_C_gt
        (trainer.py:130-198).  bbox: anything with .center [3], .R [3,3], .extent [3] (the reference passes an
        open3d OrientedBoundingBox).  draws: optional [n_hit, n_bins] tensor replacing the torch.rand of
        stratified_bins (utils.py:371) -- tests inject the reference's draw.  Sets self.dirs_W, self.origins,
        self.z_vals, self.input_pcs (device tensors) and returns (hit_mask, near[hit], far[hit]) or
        (None, None, None) when at most one ray hits (:164-165)."""
        n_bins = 60 if self.obj_id == 0 else 20                  # :141-144
        if do_eval:
            n_bins = 150
        dev = self.device
        T_WC = torch.as_tensor(self.T_WC_gt[0], dtype=torch.float32).cpu()     # one view: every row is the same pose
        T_WO = torch.eye(4)
        T_WO[:3, :3] = torch.as_tensor(np.asarray(bbox.R))                      # :152-154
        T_WO[:3, 3] = torch.as_tensor(np.asarray(bbox.center))
        T_OC = torch.inverse(T_WO) @ T_WC                                       # :156-157
        half = torch.as_tensor(np.asarray(bbox.extent, np.float64) / 2.0).float()       # :161
        dirs_C = torch.as_tensor(self.dirs_C_gt, dtype=torch.float32).to(dev).reshape(-1, 3)
        dirs_W, near, far, hit = ops.box_rays(T_WC, T_OC, half, dirs_C)
        n_rays = int(hit.sum())
        if n_rays <= 1:
            # (nothing of an earlier call may be served by the lazy z_vals / input_pcs properties after this)
            self._bbox_samples = None
            self._z_vals = self._input_pcs = None
            return None, None, None
        self.dirs_W = dirs_W[hit]
        self.origins = T_WC[:3, 3].to(dev).expand(n_rays, 3)
        near_h, far_h = near[hit].contiguous(), far[hit].contiguous()
        u = None if draws is None else torch.as_tensor(draws).to(dev)     # None: drawn inside the kernel (seeded)
        # z_vals / input_pcs are formed on first access (properties below): the fused renderer (vmap.render_2D_syn ->
        # ops.render_fwd) never needs the [n, 149, 3] point tensor; the same (seed, draw) reproduces the same numbers
        # (the process-wide Philox call counter advances only when this call DRAWS: injected draws leave every later
        # seeded launch on the stream it would have had without this call)
        draw = (ops._next_offset() & 0x1FFFFFFF) if u is None else 0
        self._bbox_samples = dict(origin=T_WC[:3, 3], dirs_W=self.dirs_W.contiguous(), near=near_h, far=far_h, u=u,
                                  n_bins=n_bins, seed=ops._seed_of(None), draw=draw)
        self._z_vals = self._input_pcs = None
        return hit, near_h, far_h

    def _materialise_bbox_samples(self):
        b = self._bbox_samples
        if b is None:
            raise RuntimeError("z_vals / input_pcs: the last sample_points_bbox call found no ray inside the box")
        self._z_vals, self._input_pcs = ops.box_points(b["origin"], b["dirs_W"], b["near"], b["far"], b["u"], b["n_bins"],
                                                       seed=b["seed"], draw=b["draw"])

    @property
    def z_vals(self):                       # trainer.py:175
        if getattr(self, "_z_vals", None) is None:
            self._materialise_bbox_samples()
        return self._z_vals

    @z_vals.setter
    def z_vals(self, v):
        self._z_vals = v

    @property
    def input_pcs(self):                    # trainer.py:176
        if getattr(self, "_input_pcs", None) is None:
            self._materialise_bbox_samples()
        return self._input_pcs

    @input_pcs.setter
    def input_pcs(self, v):
        self._input_pcs = v

    def meshing(self, *a, **k):
        raise NotImplementedError("marching cubes / open3d meshing (trainer.py:46-103, vis.py) is outside the "
                                  "accelerated path; evaluate the grid with eval_points(render_rays.make_3D_grid(...))")
