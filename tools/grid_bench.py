"""Dense-grid evaluation for meshing (trainer.py:46-128: make_3D_grid + eval_points at cfg grid_dim = 128, plus a
256^3 occupancy / colour grid without the 512-d features).  Run on the GPU box."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openobj_amd import cfg as ocfg, ops, trainer

dev = "cuda:0"


def grid(dim):
    lin = torch.linspace(-1, 1, dim, device=dev)
    return torch.stack(torch.meshgrid(lin, lin, lin, indexing="ij"), -1).reshape(-1, 3)


def timed(fn):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return time.perf_counter() - t0


for name, hidden in (("object network (hidden 32)", 32), ("background network (hidden 128)", 128)):
    c = ocfg.Config(ocfg.replica_room0_config(train_device=dev))
    c.obj_id = 1
    c.hidden_feature_size = hidden
    t = trainer.Trainer(c)
    p = grid(128)
    dt = timed(lambda: t.eval_points(p))
    print("%s: Trainer.eval_points on 128^3 (occupancy, colour, 512-d features): %.1f ms = %.1f M points/s"
          % (name, 1e3 * dt, p.shape[0] / dt / 1e6), flush=True)
    p = grid(256).reshape(1, -1, 3).contiguous()
    dt = timed(lambda: ops.eval_points(t.arena, p))
    print("%s: occupancy + colour on 256^3: %.1f ms = %.1f M points/s" % (name, 1e3 * dt, p.shape[1] / dt / 1e6), flush=True)
