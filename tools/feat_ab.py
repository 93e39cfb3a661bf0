"""Diagnostic: the feature-loss iteration at several shapes, default library vs a variant (OBJNERF_LIB), ms per step."""
import os, sys, time
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import torch
from openobj_amd import ops, synthetic, init as obj_init
dev = torch.device("cuda:0")
for (K, R, n1, n2) in [(50, 120, 1, 9), (50, 1024, 8, 24), (50, 4096, 16, 48), (15, 4096, 16, 48)]:
    arena = ops.ParamArena(K, ops.NetShape(), dev)
    arena.load_stacked(obj_init.init_stacked(K, 32, 512, seed=1))
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
    b = synthetic.random_batch(K, R, n1, n2, seed=1, feat_dim=512)
    batch = {k: torch.from_numpy(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels", "gt_feat"]}
    for _ in range(5):
        ops.train_step(arena, ws, batch, with_feat=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 30
    for _ in range(n):
        ops.train_step(arena, ws, batch, with_feat=True)
    torch.cuda.synchronize()
    print(os.environ.get("OBJNERF_LIB", "default")[-16:], (K, R, n1 + n2), "%.3f ms" % ((time.perf_counter() - t0) / n * 1e3))
