"""Two-rank check of the object-sharded mapping loop on ONE GPU (gloo carries the collectives, both ranks compute on
cuda:0):   python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
               tools/mapping_dist_check.py
Checks: foreground objects are dealt to their owners, the replicated background network stays bit-identical on both
ranks while it trains on disjoint ray shares, losses fall, checkpoints are written once."""
import os
import sys

import torch
import torch.distributed as dist

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from openobj_amd import cfg as ocfg, dataset as ods, mapping      # noqa: E402
from tests import scene_files as SF                               # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
scene = "/tmp/mapping_dist_scene"
if rank == 0:
    SF.write_scene(scene, "Replica", n_frames=50)
dist.barrier()
c = ocfg.Config(ocfg.replica_room0_config(train_device="cuda:0", **{
    "dataset.path": scene, "dataset.format": "Replica", "trainer.part_mode": 0, "camera.w": SF.W, "camera.h": SF.H,
    "camera.fx": SF.FX, "camera.fy": SF.FY, "camera.cx": SF.CX, "camera.cy": SF.CY, "render.iters_per_frame": 30,
    "render.n_per_optim_bg": 240}))
torch.manual_seed(1234)
m = mapping.IncrementalMapper(c)
hist = []
m.run(ods.init_loader(c, multi_worker=False), on_frame=lambda f, l: hist.append(l))
want = [4] if rank == 0 else [7]
assert list(m.obj_dict) == want and m.remote_ids == ({7} if rank == 0 else {4}), (rank, list(m.obj_dict), m.remote_ids)
assert m.n_foreground == 2 and m.scene_bg is not None
# background replicas: bit-identical parameters after 150 all-reduced steps on disjoint ray shares
p = m.scene_bg.trainer.arena.params.clone()
both = [torch.empty_like(p) for _ in range(world)]
dist.all_gather(both, p)
assert torch.equal(both[0], both[1]), float((both[0] - both[1]).abs().max())
# each rank drew 12 of the 24 background rays per keyframe
assert hist[-1]["bg"][0].shape == (1, 4)
tot = lambda terms: float(sum((t[..., 0] + 5 * t[..., 1] + 10 * t[..., 2]).sum() for t in terms) / len(terms))
assert tot(hist[-1]["obj"]) < 0.5 * tot(hist[0]["obj"]), (tot(hist[0]["obj"]), tot(hist[-1]["obj"]))
assert tot(hist[-1]["bg"]) < 0.7 * tot(hist[0]["bg"]), (tot(hist[0]["bg"]), tot(hist[-1]["bg"]))
log = "/tmp/mapping_dist_log"
m.save_checkpoints(log)
dist.barrier()
if rank == 0:
    assert sorted(os.listdir(os.path.join(log, "ckpt"))) == ["0", "4", "7"], os.listdir(os.path.join(log, "ckpt"))
print("rank %d OK: objects %s, obj loss %.3f -> %.3f, bg loss %.3f -> %.3f" % (
    rank, list(m.obj_dict), tot(hist[0]["obj"]), tot(hist[-1]["obj"]), tot(hist[0]["bg"]), tot(hist[-1]["bg"])), flush=True)
dist.destroy_process_group()
