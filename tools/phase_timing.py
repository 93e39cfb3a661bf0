"""Diagnostic: per-phase s_memtime breakdown of the fused fp32 training kernel (workgroup 0, every wave).
Build first:  tools/build_variant.sh PHASE -DPHASE_TIMING ;  run on the GPU box (--bf16: the bf16 kernel, --feat)."""
import ctypes as C, os, sys
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("OBJNERF_LIB", os.path.join(root, "openobj_amd/csrc/variants/libobjnerf_hip_PHASE.so"))
sys.path.insert(0, root)
import torch
from openobj_amd import _lib, ops, synthetic, init as obj_init

NAMES = ["load+project", "embed(sincos fwd)", "mlp fwd", "sync1", "composite", "sync2", "phaseA compute", "sync3",
         "wgrad A", "sync4", "phaseB compute", "(dB, first generation)", "sync5", "wgrad B", "sync6+phaseC stores",
         "sync7", "wgrad C", "sync8"]
dev = torch.device("cuda:0")
K, R, n1, n2 = 50, 4096, 16, 48
arena = ops.ParamArena(K, ops.NetShape(), dev)
arena.load_stacked(obj_init.init_stacked(K, 32, 512, seed=1000))
FEAT = "--feat" in sys.argv
BF16 = "--bf16" in sys.argv or "--bf16v2" in sys.argv
V2 = "--bf16v2" in sys.argv           # second-generation bf16 kernel (objnerf_train_bf16v2.hip)
if V2:
    NAMES = ["load+project", "embed + mlp fwd + heads", "sync1", "composite | wgrad(t-1)", "sync2", "bwd: d_hc, d_h4",
             "bwd: x2 chain rule", "bwd: d_h3 .. d_h1", "bwd: x1 chain rule, dB, next point"]
    if FEAT:
        NAMES += ["feat: sync a", "feat: partial fh + sync", "feat: ray term + sync", "feat: composite bwd | wgrad(t-1) w0-1"]
ws = ops.TrainWorkspace(arena, K, R, n1 + n2, FEAT)
b = synthetic.random_batch(K, R, n1, n2, seed=4242, feat_dim=512 if FEAT else 0)
batch = {k: torch.from_numpy(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if FEAT else [])}
for _ in range(3):
    ops.train_step(arena, ws, batch, with_feat=FEAT, bf16=BF16)
torch.cuda.synchronize()
out = (C.c_ulonglong * (8 * 24))()
f = ((_lib.lib().objnerf_debug_phase_bf16v2f if FEAT else _lib.lib().objnerf_debug_phase_bf16v2) if V2 else _lib.lib().objnerf_debug_phase_bf16) if BF16 else _lib.lib().objnerf_debug_phase32
f.restype = C.c_int
assert f(out) == 0
a = np.array(list(out), dtype=np.float64).reshape(8, 24)[:, :18]
tot = a.sum(1)
print("total ticks per wave:", tot.astype(np.int64))
print("%-26s" % "phase" + "".join("   w%d " % w for w in range(8)) + "   mean%")
for i, n in enumerate(NAMES):
    print("%-26s" % n + "".join("%6.1f" % (100 * a[w, i] / tot[w]) for w in range(8)) + "  %6.1f" % (100 * a[:, i].sum() / tot.sum()))
