"""vmap -- scene-object state of the reference's vmap.py that feeds the hot path: cameraInfo
(:689-720), sceneObject's keyframe buffers + append_keyframe (:29-257), the pixel / ray sampler
get_training_samples + sample_3d_points (:386-554) on objnerf_sample_rays, and the checkpoint dict
(:556-602).  3-D bounding-box estimation, point-cloud denoising and novel-view rendering
(get_bound, render_2D_syn) are not on the accelerated training path.
"""
import copy
import os
import random

import numpy as np
import torch

from . import ops, trainer


class cameraInfo:
    def __init__(self, cfg) -> None:
        self.device = cfg.data_device
        self.width, self.height = cfg.W, cfg.H
        self.fx, self.fy, self.cx, self.cy = cfg.fx, cfg.fy, cfg.cx, cfg.cy
        self.rays_dir_cache = self.get_rays_dirs()

    def get_rays_dirs(self, depth_type="z"):
        """(W,H,3) un-normalised camera rays, image stored transposed (vmap.py:701-720)."""
        if depth_type != "z":
            raise Exception("Get camera rays directions with euclidean depth not yet implemented")
        return ops.rays_dirs(self.width, self.height, float(self.fx), float(self.fy), float(self.cx),
                             float(self.cy), self.device)


RENDER_SAMPLES_PER_CHUNK = 1 << 23      # samples per launch chain of sceneObject.render_2D_syn


class sceneObject:
    """Keyframe ring buffers of one object + its Trainer (networks)."""

    def __init__(self, cfg, obj_id, rgb, depth, mask, bbox_2d, t_wc, live_frame_id, clip_feat=None,
                 caption_feat=None, defer=None) -> None:
        """defer: a list -- the frame is NOT written here; (self, slot, bbox_2d) is appended to it and the caller
        writes every object's slot of this frame in one launch (ops.ingest_frame; mask may then be None)."""
        self._defer = defer
        self.do_bg = cfg.do_bg
        self.obj_id = obj_id
        self.data_device = cfg.data_device
        self.training_device = cfg.training_device
        self.part_mode = cfg.part_mode
        self.stride = cfg.stride
        assert rgb.shape[:2] == depth.shape
        assert defer is not None or rgb.shape[:2] == mask.shape
        assert bbox_2d.shape == (4,)
        assert t_wc.shape == (4, 4,)
        if self.do_bg and self.obj_id == 0:      # separate background network (vmap.py:43-52)
            self.obj_scale = cfg.bg_scale
            self.hidden_feature_size = cfg.hidden_feature_size_bg
            self.n_bins_cam2surface = cfg.n_bins_cam2surface_bg
            self.keyframe_step = cfg.keyframe_step_bg
        else:
            self.obj_scale = cfg.obj_scale
            self.hidden_feature_size = cfg.hidden_feature_size
            self.n_bins_cam2surface = cfg.n_bins_cam2surface
            self.keyframe_step = cfg.keyframe_step
        self.frames_width, self.frames_height = rgb.shape[0], rgb.shape[1]
        self.min_bound, self.max_bound = cfg.min_depth, cfg.max_depth
        self.n_bins = cfg.n_bins
        self.n_unidir_funcs = cfg.n_unidir_funcs
        self.surface_eps, self.stop_eps = cfg.surface_eps, cfg.stop_eps
        self.n_keyframes = 1
        self.kf_pointer = None
        self.keyframe_buffer_size = cfg.keyframe_buffer_size
        self.kf_id_dict = {live_frame_id: 0}        # frame id -> slot (insertion ordered, like the bidict)
        self.kf_buffer_full = False
        self.frame_cnt = 0
        self.lastest_kf_queue = []
        self.feat_cnt = 1
        self.clip_feat, self.caption_feat = clip_feat, caption_feat
        dev = self.data_device
        self.bbox = torch.empty(self.keyframe_buffer_size, 4, device=dev)
        if defer is None:
            self.bbox[0] = bbox_2d
        self.rgb_idx, self.state_idx = slice(0, 3), slice(3, 4)
        self.rgbs_batch = torch.empty(self.keyframe_buffer_size, self.frames_width, self.frames_height, 4,
                                      dtype=torch.uint8, device=dev)
        if self.part_mode:
            self.part_down = cfg.part_down
            self.use_frame = np.zeros(self.keyframe_buffer_size)
            self.use_frame[0] = live_frame_id
        self.other_obj, self.this_obj, self.unknown_obj = 0, 1, 2
        self.semantic_id = None
        self.depth_batch = torch.empty(self.keyframe_buffer_size, self.frames_width, self.frames_height,
                                       dtype=torch.float32, device=dev)
        self.t_wc_batch = torch.empty(self.keyframe_buffer_size, 4, 4, dtype=torch.float32, device=dev)
        if defer is None:
            self.rgbs_batch[0, :, :, self.rgb_idx] = rgb
            self.rgbs_batch[0, :, :, self.state_idx] = mask[..., None]
            self.depth_batch[0] = depth
            self.t_wc_batch[0] = t_wc
        else:
            defer.append((self, 0, bbox_2d))
        trainer_cfg = copy.deepcopy(cfg)
        trainer_cfg.obj_id = self.obj_id
        trainer_cfg.hidden_feature_size = self.hidden_feature_size
        trainer_cfg.obj_scale = self.obj_scale
        self.trainer = trainer.Trainer(trainer_cfg)
        self.bbox_final = False
        self.bbox3dour = None
        self.obj_center = torch.tensor(0.0)

    # ------------------------------------------------------------------ keyframes (vmap.py:166-257)
    def _write_slot(self, slot, rgb, depth, mask, bbox_2d, t_wc, frame_id):
        if self._defer is not None:             # the caller writes all objects' slots of this frame in one launch
            self._defer.append((self, slot, bbox_2d))
            if self.part_mode:
                self.use_frame[slot] = frame_id
            return
        self.rgbs_batch[slot, :, :, self.rgb_idx] = rgb
        self.rgbs_batch[slot, :, :, self.state_idx] = mask[..., None]
        self.depth_batch[slot, ...] = depth
        self.t_wc_batch[slot, ...] = t_wc
        self.bbox[slot, ...] = bbox_2d
        if self.part_mode:
            self.use_frame[slot] = frame_id

    def _rebind(self, slot, frame_id):
        for k, v in list(self.kf_id_dict.items()):
            if v == slot:
                del self.kf_id_dict[k]
        self.kf_id_dict[frame_id] = slot

    def append_keyframe(self, rgb, depth, mask, bbox_2d, t_wc, frame_id=1, clip_feat=None, caption_feat=None):
        assert rgb.shape[:2] == depth.shape and (self._defer is not None or rgb.shape[:2] == mask.shape)
        assert bbox_2d.shape == (4,) and t_wc.shape == (4, 4,)
        assert self.n_keyframes <= self.keyframe_buffer_size - 1
        assert rgb.dtype == torch.uint8 and depth.dtype == torch.float32
        assert self._defer is not None or mask.dtype == torch.uint8
        is_kf = (self.frame_cnt % self.keyframe_step == 0) or self.n_keyframes == 1
        if self.n_keyframes == self.keyframe_buffer_size - 1:      # buffer full: overwrite the free slot
            self.kf_buffer_full = True
            if self.kf_pointer is None:
                self.kf_pointer = self.n_keyframes
            self._write_slot(self.kf_pointer, rgb, depth, mask, bbox_2d, t_wc, frame_id)
            self._rebind(self.kf_pointer, frame_id)
            if is_kf:
                self.lastest_kf_queue.append(self.kf_pointer)
                _, self.kf_pointer = self.prune_keyframe()
        elif not is_kf:                                            # replace the live (last) slot
            self._write_slot(self.n_keyframes - 1, rgb, depth, mask, bbox_2d, t_wc, frame_id)
            self._rebind(self.n_keyframes - 1, frame_id)
        else:                                                      # add a new keyframe
            self.kf_id_dict[frame_id] = self.n_keyframes
            self._write_slot(self.n_keyframes, rgb, depth, mask, bbox_2d, t_wc, frame_id)
            self.lastest_kf_queue.append(self.n_keyframes)
            self.n_keyframes += 1
        self.frame_cnt += 1
        if clip_feat is not None:
            self.clip_feat = np.vstack((self.clip_feat, clip_feat))
            self.caption_feat = np.vstack((self.caption_feat, caption_feat))
            self.feat_cnt += 1
        if len(self.lastest_kf_queue) > 2:
            self.lastest_kf_queue = self.lastest_kf_queue[-2:]

    def prune_keyframe(self):
        key, value = random.choice(list(self.kf_id_dict.items())[:-2])   # never the latest two
        return key, value

    # ------------------------------------------------------------------ sampling (vmap.py:386-554)
    def draw_keyframe_ids(self, n_frames):
        dev = self.data_device
        if self.n_keyframes > 2:      # the latest two keyframes are always included, LAST (vmap.py:390-401)
            ids = torch.randint(low=0, high=self.n_keyframes, size=(n_frames - 2,), dtype=torch.long, device=dev)
            return torch.cat([ids, torch.tensor(self.lastest_kf_queue[-2:], dtype=torch.long, device=dev)])
        return torch.randint(low=0, high=self.n_keyframes, size=(n_frames,), dtype=torch.long, device=dev)

    def kf_meta(self):
        """[n_keyframes, slot of the second-latest keyframe, slot of the latest (-1, -1 while n_keyframes <= 2), object
        id]: what the seeded sampler needs to choose keyframes as draw_keyframe_ids does, and the object's
        random-stream id."""
        return [self.n_keyframes] + (list(self.lastest_kf_queue[-2:]) if self.n_keyframes > 2 else [-1, -1]) + \
               [int(self.obj_id) & 0x7FFFFFFF]

    def _partfeat_args(self, global_partfeat):
        """The part-feature gather of vmap.py:437-452 as arguments of the sampler launch (ops._partfeat_fields)."""
        if not (self.part_mode and global_partfeat is not None):
            return None
        return (global_partfeat, self.use_frame, self.stride, self.part_down)

    def get_training_samples(self, n_frames, n_samples, cached_rays_dir, global_partfeat=None, draws=None, seed=None,
                             compact=False):
        """Returns the reference's 7-tuple (rgb u8, depth, valid_depth_mask[flat], obj_labels[flat] u8,
        input_pcs, sampled_z, sampled_partfeat).  `draws` = dict(kf_ids, u_w, u_h, u, g) injects the reference's
        random numbers (exact parity); otherwise every draw is generated inside the sampler kernels (Philox under
        `seed`, default torch.initial_seed(), and a per-call counter) and no random tensor exists.
        compact (seeded form only): input_pcs is returned as the pair (origins [n,3], dirs [n,3]) -- the point tensor
        is never written, ops.train_step forms the points in registers."""
        dev = self.data_device
        N, M = self.n_bins_cam2surface, self.n_bins
        n = n_frames * n_samples
        if draws is None:
            meta = torch.tensor(self.kf_meta(), dtype=torch.int32).to(dev)
            o = ops.sample_rays_seeded(self.keyframe_store(), self.keyframe_buffer_size, self.frames_width,
                                       self.frames_height, cached_rays_dir, meta, n_frames, n_samples, N, M,
                                       self.surface_eps, self.stop_eps, float(self.min_bound), float(self.obj_center),
                                       seed=seed, want_pts=not compact, partfeat=self._partfeat_args(global_partfeat))
            partfeat = o["partfeat"]
            if partfeat is not None:
                partfeat = partfeat.reshape(n_frames, n_samples, -1)
            S = N + M
            pcs = (o["origins"], o["dirs"]) if compact else o["pts"].reshape(n_frames, n_samples, S, 3)
            return (o["rgb"].reshape(n_frames, n_samples, 3), o["depth"].reshape(n_frames, n_samples), o["valid"],
                    o["labels"], pcs, o["z"].reshape(n_frames, n_samples, S), partfeat)
        pf = self._partfeat_args(global_partfeat)                   # vmap.py:437-452: gathered by the same launch
        r = ops.sample_rays(
            self.rgbs_batch, self.depth_batch, self.t_wc_batch, self.bbox, cached_rays_dir, draws["kf_ids"],
            draws["u_w"], draws["u_h"], draws["u"], draws["g"], N, M, self.surface_eps, self.stop_eps,
            float(self.min_bound), float(self.obj_center), partfeat=pf)
        return r if pf is not None else r + (None,)

    def sample_3d_points(self, sampled_rgbs, sampled_depth, origins, dirs_w, sampled_partfeat=None, draws=None,
                         seed=None):
        """vmap.py:456-554 as a callable of its own (get_training_samples above runs gather + placement as one launch
        chain): sampled_rgbs u8 [n_frames, n_px, 4] (rgb + state), sampled_depth [n_frames, n_px], origins
        [n_frames, 3], dirs_w [n_frames, n_px, 3] -> the reference's 7-tuple (rgb u8, depth, valid_depth_mask[flat],
        obj_labels[flat] u8, input_pcs, sampled_z, sampled_partfeat); sampled_partfeat is returned as given (:554).
        draws = dict(u [n, N + M], g [n, M]) injects the reference's torch.rand / normal_ numbers in ray order (exact
        parity); otherwise the placement kernel draws them itself (Philox under `seed`)."""
        u = draws["u"] if draws is not None else None
        g = draws["g"] if draws is not None else None
        r = ops.sample_points(sampled_rgbs, sampled_depth, origins, dirs_w, self.n_bins_cam2surface, self.n_bins,
                              self.surface_eps, self.stop_eps, float(self.min_bound), float(self.obj_center), u=u, g=g,
                              seed=seed, obj_index=int(self.obj_id) & 0x7FFFFFFF)
        return r + (sampled_partfeat,)

    def keyframe_store(self):
        """The four device tensors the sampler reads (fixed addresses for the life of the object)."""
        return self.rgbs_batch, self.depth_batch, self.t_wc_batch, self.bbox

    def set_semantic(self, semantic_id):                                  # vmap.py:284-285
        self.semantic_id = semantic_id

    # ------------------------------------------------------------------ checkpoints (vmap.py:556-602)
    def get_bound(self, intrinsic_open3d=None, final=True):
        """The object's oriented 3-D box (vmap.py:259-384 builds it from the keyframe point cloud with open3d,
        outside the accelerated path): returns (None, self.bbox3dour), which the caller must have set."""
        if self.bbox3dour is None:
            raise RuntimeError("sceneObject.bbox3dour is not set: assign an oriented box (.center, .R, .extent)")
        return None, self.bbox3dour

    def render_2D_syn(self, T_WC, intrinsic_open3d, cached_rays_dir, T_WO=None, chunk_size=1000, do_fine=True,
                      obj_mask=None, render_part=False, draws=None):
        """Render this object from camera pose T_WC [4,4] inside its oriented box (vmap.py:604-685).
        Returns (obj_mask [W,H] bool ndarray, depth [n], colour [n,3] uint8, feature [n,C] | None) for the
        n pixels that hit the box, terminate inside it (near <= depth <= far) and reach opacity 0.9 -- or
        (None, None, None).  The whole view is ONE launch chain on the device: box sampler -> fused PE + MLP ->
        compositing (depth, rgb, opacity and the H-wide feature hidden) -> the linear 512-d head applied to the
        composited hidden (exact, the head is linear).  Rays are processed in chunks of RENDER_SAMPLES_PER_CHUNK
        samples; the reference's chunk_size (points per network call) is accepted for signature parity only."""
        tr = self.trainer
        W, H = tr.W_vis, tr.H_vis
        _, bbox = self.get_bound(intrinsic_open3d, final=True)
        dev = self.training_device
        if obj_mask is None:
            obj_mask = np.ones([W, H], dtype=bool)
        mask_t = torch.as_tensor(obj_mask).to(dev)
        idx = mask_t.nonzero(as_tuple=True)
        tr.T_WC_gt = torch.as_tensor(T_WC, dtype=torch.float32).unsqueeze(0)
        tr.dirs_C_gt = cached_rays_dir.to(dev)[idx[0], idx[1]]
        obj_hit, obj_near, obj_far = tr.sample_points_bbox(bbox, do_eval=True, draws=draws)
        if obj_hit is None:
            return None, None, None
        b = tr._bbox_samples
        n_pts, S = int(obj_near.shape[0]), b["n_bins"] - 1
        if n_pts <= 1:
            print("too few hits")
            return None, None, None
        tr.arena.scale.fill_(float(tr.obj_scale))
        with torch.no_grad():
            if tr.hidden_feature_size == 32:
                # ONE launch: mid-points -> embedding -> network -> compositing with a lane per ray (objnerf_render_fwd);
                # no point tensor, no per-sample alpha / colour / feature tensors
                # (trainer.render_bf16 = True: the opt-in bf16-operand arithmetic, ~3x faster, not the reference's fp32)
                o = ops.render_fwd(tr.arena, b["origin"], b["dirs_W"], b["near"], b["far"], b["u"], b["n_bins"],
                                   seed=b["seed"], draw=b["draw"], want_hfeat=render_part,
                                   bf16=bool(getattr(tr, "render_bf16", False)))
                depth, opacity, rgb, fh = o["depth"], o["opacity"], o["rgb"], o["vals"]
            else:
                # wider networks (the background, hidden 128): layer-wise evaluation in ray chunks that bound the live
                # activations (the H-wide feature hidden is 4 H bytes per sample)
                rays_per_chunk = max(1, RENDER_SAMPLES_PER_CHUNK // S)
                depth = torch.empty(n_pts, device=dev)
                opacity = torch.empty(n_pts, device=dev)
                rgb = torch.empty(n_pts, 3, device=dev)
                fh = torch.empty(n_pts, tr.hidden_feature_size, device=dev) if render_part else None
                for r0 in range(0, n_pts, rays_per_chunk):
                    r1 = min(n_pts, r0 + rays_per_chunk)
                    alpha, color, hfeat, _ = ops.eval_points(tr.arena, tr.input_pcs[r0:r1].reshape(1, -1, 3),
                                                             want_hfeat=render_part)
                    o = ops.composite(alpha.reshape(r1 - r0, S), color.reshape(r1 - r0, S, 3), tr.z_vals[r0:r1],
                                      vals=hfeat.reshape(r1 - r0, S, -1) if render_part else None)
                    depth[r0:r1], opacity[r0:r1], rgb[r0:r1] = o["depth"], o["opacity"], o["rgb"]
                    if render_part:
                        fh[r0:r1] = o["vals"]
            out = dict(rgb=rgb, vals=fh)
            bad = (depth < obj_near) | (depth > obj_far) | (opacity < 0.9)          # :665,672
            keep = ~bad
            render_color = (out["rgb"] * 255).to(torch.uint8)                       # :671 (truncation)
            feat = None
            if render_part:
                # the 512-d head only for the rays that survive the masks (it is linear: F = W_of fh + b_of O)
                fh_k, op_k = out["vals"][keep].contiguous(), opacity[keep].contiguous()
                nk = fh_k.shape[0]
                feat = (ops.feature_head(tr.arena, fh_k.reshape(1, nk, -1), op_k.reshape(1, nk))[0].cpu().numpy()
                        if nk > 0 else np.zeros((0, tr.clip_point_feature_size), np.float32))
            sel = torch.zeros(idx[0].shape[0], dtype=torch.bool, device=dev)
            hit_idx = obj_hit.nonzero(as_tuple=True)[0]
            sel[hit_idx[keep]] = True                                               # :627-629 and :681-683
            mask_out = torch.zeros_like(mask_t)
            mask_out[idx[0][sel], idx[1][sel]] = True
        return (mask_out.cpu().numpy(), depth[keep].cpu().numpy(), render_color[keep].cpu().numpy(), feat,)

    def save_checkpoints(self, path, epoch):
        torch.save({
            "epoch": epoch,
            "FC_state_dict": self.trainer.fc_occ_map.state_dict(),
            "PE_state_dict": self.trainer.pe.state_dict(),
            "obj_id": self.obj_id,
            "bbox": self.bbox3dour,
            "obj_scale": self.trainer.obj_scale,
            "clip_feat": self.clip_feat,
            "caption_feat": self.caption_feat,
            "semantic_id": self.semantic_id,
        }, path + "/obj_" + str(self.obj_id) + ".pth")

    def load_checkpoints(self, ckpt_file):
        if not os.path.exists(ckpt_file):
            print("ckpt not exist ", ckpt_file)
            return
        checkpoint = torch.load(ckpt_file, weights_only=False)
        with torch.no_grad():
            for k, v in checkpoint["FC_state_dict"].items():     # copy INTO the arena views
                self.trainer.fc_occ_map.state_dict()[k].copy_(v)
            self.trainer.pe.B_layer.weight.copy_(checkpoint["PE_state_dict"]["B_layer.weight"])
        self.obj_id = checkpoint["obj_id"]
        self.bbox3dour = checkpoint["bbox"]
        self.trainer.obj_scale = checkpoint["obj_scale"]
        if "clip_feat" not in checkpoint.keys():
            print("no clip_feat for this obj:", self.obj_id)
            return False
        self.clip_feat = checkpoint["clip_feat"]
        self.caption_feat = checkpoint["caption_feat"]
        self.semantic_id = checkpoint["semantic_id"]
        self.bbox_final = True
        return True


class StackedSampler:
    """get_training_samples for a LIST of objects in one launch chain (train.py:316-330 + the stacking of
    train.py:368-388): one set of random draws for all objects, objnerf_sample_rays_stacked, outputs already in the
    [K, n, ...] layout of the training step.  The objects must share the sampler configuration (every foreground
    object does, vmap.py:53-62).  Rebuild it when the list of objects changes (the descriptor table holds their
    keyframe-store addresses)."""

    def __init__(self, objs):
        self.objs = list(objs)
        o = self.objs[0]
        for x in self.objs:
            assert (x.n_bins_cam2surface, x.n_bins, x.keyframe_buffer_size, x.frames_width, x.frames_height) == \
                   (o.n_bins_cam2surface, o.n_bins, o.keyframe_buffer_size, o.frames_width, o.frames_height)
        self.table = ops.keyframe_table([x.keyframe_store() for x in self.objs])

    def draw(self, n_frames, n_samples):
        """kf_ids as draw_keyframe_ids (uniform over the stored keyframes, the latest two always included, LAST),
        pixel / depth draws as get_training_samples -- for all objects at once."""
        o, K = self.objs[0], len(self.objs)
        dev = o.data_device
        N, M = o.n_bins_cam2surface, o.n_bins
        n = n_frames * n_samples
        nk = torch.tensor([x.n_keyframes for x in self.objs], dtype=torch.float32)
        last = torch.tensor([(x.lastest_kf_queue[-2:] if x.n_keyframes > 2 else [-1, -1]) for x in self.objs],
                            dtype=torch.int64)
        meta = torch.cat([nk[:, None], last.float()], dim=1).to(dev)            # one small host -> device copy
        kf = torch.floor(torch.rand(K, n_frames, device=dev) * meta[:, :1]).long()
        kf = torch.minimum(kf, (meta[:, :1] - 1).long())                        # (rand() < 1, guard the rounding)
        if n_frames >= 2:
            tail = meta[:, 1:].long()
            kf[:, -2:] = torch.where(tail >= 0, tail, kf[:, -2:])
        return dict(kf_ids=kf, u_w=torch.rand(K, n_frames, n_samples, device=dev),
                    u_h=torch.rand(K, n_frames, n_samples, device=dev), u=torch.rand(K, n, N + M, device=dev),
                    g=torch.empty(K, n, M, device=dev).normal_(mean=0., std=o.surface_eps / 3.))

    def sample(self, n_frames, n_samples, cached_rays_dir, global_partfeat=None, draws=None, seed=None, compact=False):
        """-> (rgb u8 [K,n,3], depth [K,n], valid [K,n], labels u8 [K,n], pts [K,n,S,3], z [K,n,S], partfeat | None)
        draws: the injected random numbers of draw() (exact parity with the reference's generator); default: seeded,
        generated inside the kernels (see sceneObject.get_training_samples).  compact (seeded only): `pts` is the
        pair (origins [K,n,3], dirs [K,n,3])."""
        o = self.objs[0]
        if draws is None:
            dev = o.data_device
            meta = torch.tensor([x.kf_meta() for x in self.objs], dtype=torch.int32).to(dev)   # one small H2D copy
            r = ops.sample_rays_seeded(self.table, o.keyframe_buffer_size, o.frames_width, o.frames_height,
                                       cached_rays_dir, meta, n_frames, n_samples, o.n_bins_cam2surface, o.n_bins,
                                       o.surface_eps, o.stop_eps, float(o.min_bound), float(o.obj_center), seed=seed,
                                       want_pts=not compact, partfeat=self._partfeat_args(global_partfeat))
            pcs = (r["origins"], r["dirs"]) if compact else r["pts"]
            return r["rgb"], r["depth"], r["valid"], r["labels"], pcs, r["z"], r["partfeat"]
        pf = self._partfeat_args(global_partfeat)                              # vmap.py:437-452, all objects at once
        r = ops.sample_rays_stacked(
            self.table, o.keyframe_buffer_size, o.frames_width, o.frames_height, cached_rays_dir, draws["kf_ids"],
            draws["u_w"], draws["u_h"], draws["u"], draws["g"], o.n_bins_cam2surface, o.n_bins, o.surface_eps,
            o.stop_eps, float(o.min_bound), float(o.obj_center), partfeat=pf)
        return r if pf is not None else r + (None,)

    def _partfeat_args(self, global_partfeat):
        o = self.objs[0]
        if not (o.part_mode and global_partfeat is not None):
            return None
        return (global_partfeat, np.stack([x.use_frame for x in self.objs]), o.stride, o.part_down)
