"""The incremental mapping loop of the reference's train.py:155-541 around the HIP training iteration
(frame ingestion either side of the hot path), for offline datasets
(`cfg.live_mode` False) and `training_strategy == "hip"`:

    for every frame:   ingest (state maps, new sceneObjects / new keyframes)          train.py:172-256
                       restack the object networks when an object appeared            train.py:272-276
                       draw the frame's sample pools (objects + background)           train.py:297-392
                       n_iter_per_frame fused iterations + background steps           train.py:394-474
                       copy the stacked parameters back                               train.py:478-485
    every n_vis_iter frames (optional): semantic labels from the accumulated CLIP / caption features
    and the checkpoint files                                                          train.py:489-541

Under `torch.distributed` (one process per GPU) the map is OBJECT-SHARDED like the training step:
every rank reads every frame, foreground objects are dealt round-robin to the ranks in order of first appearance
and live only on their owner (keyframes, networks, optimiser state, checkpoints); the background network is
replicated, each rank trains it on its share of the frame's background rays and the gradient is SUM all-reduced
(`train.BackgroundLoop`).  The only other exchange is the pair of early-return flags per iteration.

Everything numerical runs in libobjnerf_hip.so (sampler, fused iteration, AdamW); this file is bookkeeping.
The class-name text features (CLIP ViT-B/32 and SBERT encoders in the reference, train.py:108-147) are inputs
here: pass `class_clipfeat` / `class_capfeat` arrays to `assign_semantics`.  Visualisation, meshing and the live
ROS mode are outside the path.
"""
import contextlib
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import dist as odist
from . import ops
from . import train as otrain
from .vmap import StackedSampler, cameraInfo, sceneObject


def get_majority_cluster_mean(vectors, eps, min_samples):
    """utils.py:138-155: DBSCAN the per-frame features, mean of the most populated cluster."""
    from sklearn.cluster import DBSCAN
    labels = DBSCAN(eps=eps, min_samples=min_samples).fit_predict(vectors)
    uniq, counts = np.unique(labels, return_counts=True)
    return np.mean(vectors[labels == uniq[np.argmax(counts)]], axis=0)


def _to_dev(x, dev, dtype=None):
    t = x if torch.is_tensor(x) else torch.as_tensor(np.asarray(x))
    return t.to(device=dev, dtype=dtype) if dtype is not None else t.to(dev)


class IncrementalMapper:
    def __init__(self, cfg, bf16: bool = False, group=None):
        self.cfg = cfg
        self.bf16 = bf16
        self.group = group
        self.rank, self.world = odist.rank_world(group)
        if self.world > 1:                      # every rank must draw DIFFERENT pixels of the shared background
            torch.manual_seed(torch.initial_seed() + 7919 * self.rank)
        self.remote_ids = set()                 # foreground objects that live on another rank
        self.n_foreground = 0                   # foreground objects seen so far (identical on every rank)
        self.cam_info = cameraInfo(cfg)
        self.obj_dict: Dict[int, sceneObject] = {}      # foreground objects (the stacked networks), creation order
        self.vis_dict: Dict[int, sceneObject] = {}      # + the background object
        self.scene_bg: Optional[sceneObject] = None
        self.global_partfeat: Optional[torch.Tensor] = None
        self._partfeat_buf: Optional[torch.Tensor] = None
        self.loop: Optional[otrain.HipTrainLoop] = None
        self.bg_loop: Optional[otrain.BackgroundLoop] = None
        self._restack = False
        self._sampler: Optional[StackedSampler] = None
        self._sampler_ids = ()
        self._side = None
        self.last_twc = None
        self.last_frame_id = None

    # ------------------------------------------------------------------ train.py:172-256
    def ingest(self, sample, frame_id: int) -> List[int]:
        """Append one dataset sample to the map.  Returns the ids of the objects created by this frame."""
        cfg = self.cfg
        dev = cfg.data_device
        rgb = _to_dev(sample["image"], dev, torch.uint8)
        depth = _to_dev(sample["depth"], dev, torch.float32)
        twc = _to_dev(sample["T"], dev, torch.float32)
        bbox_dict, obj_clip, obj_cap = sample["bbox_dict"], sample["obj_clip"], sample["obj_cap"]
        live_frame_id = int(sample["frame_id"]) if "frame_id" in sample else frame_id
        if cfg.part_mode:
            # train.py:187-191 grows the [frames, W', H', C] tensor with torch.cat every frame (67 MB per frame at
            # 240 x 136 x 512: quadratic copying); here the buffer doubles its capacity and the frame is copied once
            part = _to_dev(sample["part_feat"], dev, torch.float32)
            cnt = 0 if self.global_partfeat is None else self.global_partfeat.shape[0]
            if self._partfeat_buf is None or cnt == self._partfeat_buf.shape[0]:
                grown = torch.empty((max(8, 2 * cnt),) + tuple(part.shape), dtype=torch.float32, device=dev)
                if cnt:
                    grown[:cnt] = self._partfeat_buf[:cnt]
                self._partfeat_buf = grown
            self._partfeat_buf[cnt] = part
            self.global_partfeat = self._partfeat_buf[:cnt + 1]
        inst = _to_dev(sample["obj"], dev, torch.int32)
        created = []
        writes = []                     # (object, slot, 2-D box): every slot of this frame is written in ONE launch
        for obj_id in torch.unique(inst).tolist():
            if obj_id == -1:
                continue
            obj_id = int(obj_id)
            if obj_id not in bbox_dict:
                continue        # (a label without a box cannot be sampled; the reference would raise KeyError)
            if obj_id in self.remote_ids:
                continue
            bbox = torch.as_tensor(np.asarray(bbox_dict[obj_id])).float()       # host; the kernel takes it by value
            clip_feat, cap_feat = _first(obj_clip[obj_id]), obj_cap[obj_id]
            if obj_id in self.vis_dict:
                so = self.vis_dict[obj_id]
                so._defer = writes
                so.append_keyframe(rgb, depth, None, bbox, twc, live_frame_id, clip_feat=clip_feat, caption_feat=cap_feat)
                continue
            is_bg = cfg.do_bg and obj_id == 0
            if not is_bg:
                if self.n_foreground >= cfg.max_n_models:
                    continue                                              # "models full" (train.py:232-234)
                owner = self.n_foreground % self.world
                self.n_foreground += 1
                if owner != self.rank:
                    self.remote_ids.add(obj_id)
                    continue
            so = sceneObject(cfg, obj_id, rgb, depth, None, bbox, twc, live_frame_id, clip_feat=clip_feat,
                             caption_feat=cap_feat, defer=writes)
            if is_bg:
                self.scene_bg = so
                odist.broadcast_(so.trainer.arena.params, 0, self.group)      # identical replicas
                self.bg_loop = otrain.BackgroundLoop(cfg, so.trainer, with_feat=bool(cfg.part_mode), group=self.group,
                                                     bf16=self.bf16)
            else:
                self.obj_dict[obj_id] = so
                self._restack = True
            self.vis_dict[obj_id] = so
            created.append(obj_id)
        if writes:
            # state map (1 this object / 2 unknown / 0 other, train.py:201-203) + rgb, depth, pose, box -> the slots
            ops.ingest_frame(rgb, depth, inst, twc, [(so.keyframe_store(), slot, so.obj_id, box.tolist())
                                                     for so, slot, box in writes])
            for so, _, _ in writes:
                so._defer = None
        self.last_twc, self.last_frame_id = twc, live_frame_id
        return created

    # ------------------------------------------------------------------ train.py:272-276
    def _ensure_stack(self):
        if not self.obj_dict:
            return
        if self._restack or self.loop is None:
            if self.loop is not None:
                self.loop.copy_back()
            self.loop = otrain.HipTrainLoop(self.cfg, [o.trainer for o in self.obj_dict.values()],
                                            with_feat=bool(self.cfg.part_mode), bf16=self.bf16)
            self._restack = False

    # ------------------------------------------------------------------ train.py:297-392
    def _pool_of(self, so: sceneObject, n_frames: int, n_samples: int):
        # seeded draws generated inside the sampler kernels; origins + directions instead of the point tensor
        rgb, depth, _valid, labels, (origins, dirs), z, feat = so.get_training_samples(
            n_frames, n_samples, self.cam_info.rays_dir_cache, self.global_partfeat, compact=True)
        n = n_frames * n_samples
        tdev = self.cfg.training_device
        pool = {"origins": origins.to(tdev), "dirs": dirs.to(tdev),
                "z": z.reshape(n, z.shape[-1]).to(tdev),
                "gt_depth": depth.reshape(n).to(tdev),
                "gt_rgb": rgb.reshape(n, 3).to(tdev).float() / 255.0,
                "labels": labels.reshape(n).to(tdev)}
        if self.cfg.part_mode:
            pool["gt_feat"] = feat.reshape(n, feat.shape[-1]).to(tdev).float()
        return pool

    def sample_pools(self):
        """-> (stacked object pool [K, n_iter*n_per_optim, ...], background pool [1, n_iter*n_per_optim_bg, ...] | None)"""
        cfg = self.cfg
        bg_pool = None
        if cfg.do_bg and self.scene_bg is not None:
            lo, hi = odist.shard_rays(cfg.n_samples_per_frame_bg, self.world, self.rank)     # this rank's share
            bp = self._pool_of(self.scene_bg, cfg.n_iter_per_frame * cfg.win_size_bg, hi - lo)
            bg_pool = {k: v[None] for k, v in bp.items()}
        if not self.obj_dict:
            assert self.world > 1, "no foreground object in the map yet"  # train.py:366
            return None, bg_pool
        # every foreground object in ONE launch chain, outputs already stacked (train.py:316-330,368-388)
        if self._sampler_ids != tuple(self.obj_dict):
            self._sampler = StackedSampler(self.obj_dict.values())
            self._sampler_ids = tuple(self.obj_dict)
        rgb, depth, _valid, labels, (origins, dirs), z, feat = self._sampler.sample(
            cfg.n_iter_per_frame * cfg.win_size, cfg.n_samples_per_frame, self.cam_info.rays_dir_cache,
            self.global_partfeat, compact=True)
        tdev = cfg.training_device
        pool = {"origins": origins.to(tdev), "dirs": dirs.to(tdev), "z": z.to(tdev), "gt_depth": depth.to(tdev),
                "gt_rgb": rgb.to(tdev).float() / 255.0, "labels": labels.to(tdev)}
        if cfg.part_mode:
            pool["gt_feat"] = feat.to(tdev).float()
        return pool, bg_pool

    # ------------------------------------------------------------------ train.py:394-485
    def train_frame(self):
        """One frame's optimisation.  Returns {"obj": [n_iter x [K,4] loss terms], "bg": [n_iter x [1,4]]}."""
        cfg = self.cfg
        self._ensure_stack()
        pool, bg_pool = self.sample_pools()
        out = {"obj": [], "bg": []}
        npo = cfg.n_per_optim
        npo_bg = bg_pool["z"].shape[1] // cfg.n_iter_per_frame if bg_pool is not None else 0
        if pool is not None:
            # iteration i trains on rays [i npo, (i+1) npo) of every object (train.py:396-404): re-lay the pool ONCE
            # as [n_iter, K, npo, ...] so that each iteration's batch is a contiguous view (no per-iteration copies)
            pool = {k: v.reshape(v.shape[0], cfg.n_iter_per_frame, npo, *v.shape[2:]).transpose(0, 1).contiguous()
                    for k, v in pool.items()}
        if bg_pool is not None:                   # the same for the background rays ([1, n_iter * npo_bg, ...] -> views)
            bg_pool = {k: v.reshape(v.shape[0], cfg.n_iter_per_frame, npo_bg, *v.shape[2:]).transpose(0, 1).contiguous()
                       for k, v in bg_pool.items()}
        sharded = odist._active(self.group)
        # The object stack and the background network are independent chains (own parameters, optimiser state and
        # batches).  On one GPU the background steps run on a second stream beside the fused object kernel; under
        # object sharding train.ShardedIteration orders them around the iteration's two collectives (the background
        # gradient's all-reduce is in flight under the object kernel).
        side = None
        if bg_pool is not None and pool is not None and not sharded and torch.device(cfg.training_device).type == "cuda":
            if self._side is None:
                self._side = torch.cuda.Stream(device=cfg.training_device)
            side = self._side
            side.wait_stream(torch.cuda.current_stream(cfg.training_device))
        sharded_it = pre = None
        if sharded:
            # the frame's pools are resident and complete: ONE pre-step exchange for all n_iter iterations (early-return
            # flags of render_rays.py:89-94 and the background's global mask counts), then one collective per
            # iteration (the background gradient), none of them on the object kernel's critical path
            sharded_it = otrain.ShardedIteration(self.loop if pool is not None else None, self.bg_loop, self.group,
                                                 device=cfg.training_device, resident=True)
            pre = sharded_it.frame_pre(pool["labels"] if pool is not None else None,
                                       bg_pool["labels"] if bg_pool is not None else None, cfg.n_iter_per_frame)
        # un-sharded: the label statistics of ALL iterations in one launch per pool (the pools are complete before the
        # first iteration, train.py:394-404) and the loss terms of the frame in one tensor per chain -- an iteration is
        # then two launches on the object stream (fused kernel; slab reduction + AdamW) and three on the background
        # stream (forward + loss + backward; grouped weight gradients; reduction + AdamW), with no copies in between
        stats = {}
        if not sharded and getattr(self.loop, "strategy", "hip") != "forloop":
            n_it = cfg.n_iter_per_frame
            dev_t = cfg.training_device
            if pool is not None:
                K_ = pool["labels"].shape[1]
                oc = ops.label_counts(pool["labels"].reshape(n_it * K_, -1))[0].reshape(n_it, K_, 2)
                stats["oc"], stats["of"] = oc, (oc == 0).any(dim=1).to(torch.int32).contiguous()
                stats["ot"] = torch.zeros(n_it, K_, 4, device=dev_t)
            if bg_pool is not None:
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    bc = ops.label_counts(bg_pool["labels"].reshape(n_it, -1))[0].reshape(n_it, 1, 2)
                    stats["bc"], stats["bf"] = bc, (bc == 0).any(dim=1).to(torch.int32).contiguous()
                    stats["bt"] = torch.zeros(n_it, 1, 4, device=dev_t)
        for it in range(cfg.n_iter_per_frame):
            batch = {k: v[it] for k, v in pool.items()} if pool is not None else None
            bg_slice = (lambda: {k: v[it] for k, v in bg_pool.items()}) if bg_pool is not None else None
            if sharded:
                ot, bt = sharded_it.step(batch, bg_slice() if bg_slice is not None else None, pre=pre[it])
                if ot is not None:
                    out["obj"].append(ot.clone())
                if bt is not None:
                    out["bg"].append(bt.clone())
                continue
            if batch is not None:
                if "oc" in stats:
                    self.loop.step(batch, global_flags=stats["of"][it], global_counts=stats["oc"][it],
                                   loss_out=stats["ot"][it])
                else:
                    out["obj"].append(self.loop.step(batch).clone())
            if bg_slice is not None:
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    if "bc" in stats:
                        self.bg_loop.step(bg_slice(), counts=stats["bc"][it], flags=stats["bf"][it],
                                          loss_out=stats["bt"][it])
                    else:
                        out["bg"].append(self.bg_loop.step(bg_slice()).clone())   # (the slice copies run on `side` too)
        if "ot" in stats:
            out["obj"] = list(stats["ot"].unbind(0))
        if "bt" in stats:
            if side is not None:        # (allocated and written on the second stream, handed to the caller's)
                main_s = torch.cuda.current_stream(cfg.training_device)
                main_s.wait_stream(side)
                stats["bt"].record_stream(main_s)
            out["bg"] = list(stats["bt"].unbind(0))
        if side is not None:
            torch.cuda.current_stream(cfg.training_device).wait_stream(side)
        # render_rays.py:109-111 ("loss explode" -> exit): every rank must leave together, so the status is MAX-reduced
        # before anyone raises (a rank-local raise would leave the others hanging in their next collective)
        status = torch.zeros(1, dtype=torch.int32, device=cfg.training_device)
        if self.loop is not None and self.loop.ws is not None:
            status = torch.maximum(status, self.loop.ws.status)
        if self.bg_loop is not None and self.bg_loop.ws is not None:
            status = torch.maximum(status, self.bg_loop.ws.status)
        if sharded:
            odist.allreduce_max_(status, self.group)
        from .render_rays import check_status
        check_status(status)                     # bit 0 raises LossExplode; a non-finite term only warns (as the reference)
        if self.loop is not None:
            self.loop.copy_back()
        return out

    def step_frame(self, sample, frame_id: int):
        self.ingest(sample, frame_id)
        if self.n_foreground == 0:
            return None
        return self.train_frame()

    def run(self, dataloader, n_frames: Optional[int] = None, on_frame=None):
        """train.py:155-485 over a dataset.init_loader(cfg) stream."""
        for frame_id, sample in enumerate(dataloader):
            if n_frames is not None and frame_id >= n_frames:
                break
            losses = self.step_frame(sample, frame_id)
            if on_frame is not None:
                on_frame(frame_id, losses)

    # ------------------------------------------------------------------ train.py:489-541
    def assign_semantics(self, class_names: List[str], class_clipfeat: np.ndarray, class_capfeat: np.ndarray,
                         eps: float = 0.2, min_samples: int = 2) -> Dict[int, int]:
        """Object id -> class index: wall / floor / ceiling are fixed for ids 0 / 2 / 3, every other object takes the
        class whose text feature is closest to its majority-cluster caption feature (if that similarity > 0.5) or
        CLIP feature (train.py:497-524)."""
        mapping = {0: class_names.index("wall"), 2: class_names.index("floor"), 3: class_names.index("ceiling")}
        for obj_id, so in self.vis_dict.items():
            if obj_id not in (0, 2, 3):
                clip_f, cap_f = so.clip_feat, so.caption_feat
                if np.ndim(clip_f) == 2:
                    clip_f = get_majority_cluster_mean(clip_f, eps, min_samples)
                    cap_f = get_majority_cluster_mean(cap_f, eps, min_samples)
                sim_clip, sim_cap = class_clipfeat @ clip_f, class_capfeat @ cap_f
                i_clip, i_cap = int(np.argmax(sim_clip)), int(np.argmax(sim_cap))
                mapping[obj_id] = i_cap if sim_cap[i_cap] > 0.5 else i_clip
            so.set_semantic(mapping[obj_id])
        return mapping

    def save_checkpoints(self, log_dir: str, need_bound: bool = False):
        """ckpt/<obj_id>/obj_<id>.pth for every object + cam_pose/twc_frame.pth (train.py:527-541).  The reference
        refreshes each object's 3-D box first (open3d point-cloud fitting, outside this build): with need_bound the
        caller must have set sceneObject.bbox3dour."""
        for obj_id, so in self.vis_dict.items():
            if obj_id == 0 and self.scene_bg is so and self.rank != 0:
                continue                                  # the replicated background is written by rank 0
            d = os.path.join(log_dir, "ckpt", str(obj_id))
            os.makedirs(d, exist_ok=True)
            if need_bound:
                so.get_bound(None)
            so.save_checkpoints(d, self.last_frame_id)
        if self.rank != 0:
            return
        cam_dir = os.path.join(log_dir, "cam_pose")
        os.makedirs(cam_dir, exist_ok=True)
        torch.save({"twc": self.last_twc}, os.path.join(cam_dir, "twc_frame.pth"))


def _first(x):
    """obj_clip[obj_id][0] of train.py:219,226 (the stored CLIP entry is a [1, C] array)."""
    a = np.asarray(x)
    return a[0] if a.ndim >= 2 else a


def main(argv=None):
    """The reference's `python train.py --config <json> --logdir <dir>` (train.py:33-46) for the offline formats:
    map the whole dataset, write the checkpoints every cfg.n_vis_iter frames and at the end."""
    import argparse
    import shutil
    from . import cfg as ocfg
    from . import dataset
    ap = argparse.ArgumentParser(description="Incremental object-NeRF mapping on MI355X (offline dataset).")
    ap.add_argument("--logdir", default="./logs/debug", type=str)
    ap.add_argument("--config", default="./configs/Replica/config_replica_room0_vMAP.json", type=str)
    ap.add_argument("--frames", default=None, type=int, help="stop after this many frames")
    ap.add_argument("--bf16", action="store_true", help="bf16 MFMA operands (fp32 accumulation and weights)")
    ap.add_argument("--single-worker", action="store_true", help="read frames in this process (no loader workers)")
    args = ap.parse_args(argv)
    cfg = ocfg.Config(args.config)
    rank = 0
    if "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1:     # torchrun: one process per GPU
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        cfg.training_device = cfg.data_device = "cuda:%d" % local
        rank = dist.get_rank()
    if rank == 0:
        os.makedirs(args.logdir, exist_ok=True)
        shutil.copy(args.config, args.logdir)
    mapper = IncrementalMapper(cfg, bf16=args.bf16)
    loader = dataset.init_loader(cfg, multi_worker=not args.single_worker)
    n_total = len(loader) if args.frames is None else min(len(loader), args.frames)

    def on_frame(frame_id, losses):
        if losses is not None and losses["obj"]:
            t = losses["obj"][-1]
            print("frame %d: %d objects, last-iteration loss terms (depth, colour, opacity, feature) = %s"
                  % (frame_id, t.shape[0], [round(float(x), 5) for x in t.sum(0).tolist()]), flush=True)
        if cfg.if_ckpt and frame_id > 0 and (frame_id % cfg.n_vis_iter == 0 or frame_id == n_total - 1):
            mapper.save_checkpoints(args.logdir)

    mapper.run(loader, n_frames=args.frames, on_frame=on_frame)
    return mapper


if __name__ == "__main__":
    main()
