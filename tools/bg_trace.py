"""Diagnostic: the kernel sequence of ONE background-network step at the reference's native shape
(hidden 128, 1200 rays x 14 samples).  rocprofv3 --kernel-trace --stats -- python3 tools/bg_trace.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openobj_amd import cfg as ocfg, synthetic, trainer, train as otrain
dev = torch.device("cuda:0")
FEAT = "--feat" in sys.argv                                  # the 512-d feature loss on the background network too
c = ocfg.Config(ocfg.replica_room0_config(train_device="cuda:0", **{"trainer.part_mode": int(FEAT)}))
c.obj_id = 0
c.hidden_feature_size = c.hidden_feature_size_bg
c.obj_scale = c.bg_scale
t = trainer.Trainer(c)
loop = otrain.BackgroundLoop(c, t, with_feat=FEAT, bf16="--bf16" in sys.argv)
N1, N2 = (16, 48) if "--metric" in sys.argv else (5, 9)      # --metric: the bench shape (64 samples per ray)
b = synthetic.random_batch(1, 1200, N1, N2, seed=1, feat_dim=512 if FEAT else 0)
batch = {k: torch.from_numpy(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if FEAT else [])}
N = int(os.environ.get("STEPS", "50"))
for _ in range(5):
    loop.step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    loop.step(batch)
torch.cuda.synchronize()
print("bg step: %.3f ms" % (1e3 * (time.perf_counter() - t0) / N))
