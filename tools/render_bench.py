"""Novel-view rendering throughput of one object (sceneObject.render_2D_syn path): rays/s and ms per view."""
import os, sys, time, types
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openobj_amd import cfg as ocfg, trainer, vmap as ovmap, ops

dev = torch.device("cuda:0")
W, H = int(os.environ.get("RW", 1200)), int(os.environ.get("RH", 680))
part = os.environ.get("PART", "1") == "1"
c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev)))
c.obj_id = 1
c.W, c.H = W, H
torch.manual_seed(0)
t = trainer.Trainer(c)
t.render_bf16 = os.environ.get("BF16", "0") == "1"     # opt-in bf16-operand renderer (objnerf_render_bf16.hip)
with torch.no_grad():
    t.fc_occ_map.out_alpha.bias.add_(-1.0)
rays = ops.rays_dirs(W, H, 600.0, 600.0, W / 2 - 0.5, H / 2 - 0.5, dev)
box = types.SimpleNamespace(center=np.array([0.0, 0.0, 2.0]), R=np.eye(3), extent=np.array([3.0, 2.0, 1.5]))
ns = types.SimpleNamespace(trainer=t, training_device=dev, get_bound=lambda *a, **k: (None, box))
T_WC = np.eye(4, dtype=np.float32)
for it in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = ovmap.sceneObject.render_2D_syn(ns, T_WC, None, rays, render_part=part)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_hit = int(t._bbox_samples["near"].shape[0])
    print(f"view {W}x{H}: {n_hit} box rays x 149 samples, kept {res[1].shape[0]}, {dt*1e3:.1f} ms, "
          f"{n_hit/dt/1e6:.2f} M rays/s, {n_hit*149/dt/1e9:.2f} G samples/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
