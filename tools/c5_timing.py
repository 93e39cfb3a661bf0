"""Phase ticks of kernel A (fwd256_kernel built with -DOBJ256_TIMING, OBJNERF_LIB=.../libobjnerf_hip_timing.so):
one step of a few full-size hidden-256 objects without / with the feature loss."""
import sys, os
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import init as obj_init
from openobj_amd import ops, synthetic

dev = torch.device("cuda:0")
mode = sys.argv[1] if len(sys.argv) > 1 else "fp16"
K, R, n1, n2, H = 4, 8192, 32, 96, 256
for feat in (False, True):
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, H, 512, seed=3))
    b = synthetic.random_batch(K, R, n1, n2, seed=11, feat_dim=512 if feat else 0)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    batch = {k: torch.as_tensor(b[k]).to(dev) for k in keys}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, precision=mode)
    for _ in range(2):
        print(f"--- feat={feat}", flush=True)
        ops.train_step(arena, ws, batch, with_feat=feat, bf16=mode)
        torch.cuda.synchronize()
