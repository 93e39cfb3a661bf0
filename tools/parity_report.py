#!/usr/bin/env python3
"""Per-tensor distance of the HIP iteration and of the fp32 oracle from the fp64-anchored oracle (GPU box).
    python tools/parity_report.py            # prints one table per configuration
Diagnostic behind the bounds of tests/test_hip_parity.py (tests/parity_util.py)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import T
from openobj_amd import init as obj_init, ops, synthetic
from parity_util import oracle_step, maxerr

dev = torch.device("cuda:0")
CONFIGS = [  # name, K, R, n1, n2, hidden, feat, layerwise, k_chunk
    ("metric K=2 R=40", 2, 40, 16, 48, 32, False, False, 2),
    ("headline K=50 R=4096 fused", 50, 4096, 16, 48, 32, False, False, 5),
    ("K=12 R=4096 layerwise", 12, 4096, 16, 48, 32, False, True, 6),
    ("c4 share K=15 feat fused", 15, 4096, 16, 48, 32, True, False, 3),
    ("c4 share K=15 feat layerwise", 15, 4096, 16, 48, 32, True, True, 3),
    ("bg native K=1 R=1200 S=14 H=128", 1, 1200, 5, 9, 128, False, False, 1),
    ("c5 object H=256 R=8192 S=128", 1, 8192, 32, 96, 256, False, False, 1),
    ("c5 object H=256 feat", 1, 8192, 32, 96, 256, True, False, 1),
    ("long ray K=2 R=12 S=128", 2, 12, 32, 96, 32, False, False, 2),
    ("K=300 R=8 S=10", 300, 8, 1, 9, 32, False, False, 300),
    ("sweep H=256 R=64 S=128", 1, 64, 32, 96, 256, False, False, 1),
    ("sweep H=256 R=512 S=128", 1, 512, 32, 96, 256, False, False, 1),
    ("sweep H=256 R=2048 S=128", 1, 2048, 32, 96, 256, False, False, 1),
    ("sweep H=256 R=8192 S=64", 1, 8192, 16, 48, 256, False, False, 1),
    ("sweep H=128 R=8192 S=128", 1, 8192, 32, 96, 128, False, False, 1),
    ("sweep H=32 R=8192 S=128 layerwise", 1, 8192, 32, 96, 32, False, True, 1),
    ("small H=128 K=1 R=1200", 1, 1200, 5, 9, 128, False, False, 1),
    ("small H=128 K=2 R=333 feat", 2, 333, 4, 7, 128, True, False, 2),
]
only = sys.argv[1:]
for name, K, R, n1, n2, H, feat, lw, kc in CONFIGS:
    if only and not any(o in name for o in only):
        continue
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    st = obj_init.init_stacked(K, H, 512, seed=int(os.environ.get("SEED", "123")))
    if os.environ.get("SOFT"):        # density head scaled down: no ray dominates the depth term
        st = [t.clone() for t in st]
        st[8] *= 0.05
        st[9] *= 0.05
    arena.load_stacked(st)
    b = synthetic.random_batch(K, R, n1, n2, seed=321, feat_dim=512 if feat else 0)
    b["labels"][:, 0] = 1
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    batch = {k: T(b[k]).to(dev) for k in keys}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, layerwise=lw)
    ops.train_step(arena, ws, batch, with_feat=feat, layerwise=lw)
    torch.cuda.synchronize()
    scale = 5.0 if H == 128 else 2.0
    arena.scale.fill_(scale)
    ops.train_step(arena, ws, batch, with_feat=feat, layerwise=lw)
    torch.cuda.synchronize()
    o32 = oracle_step(list(st[:18]), st[18], scale, b, feat, device=dev, k_chunk=kc)
    o64 = oracle_step(list(st[:18]), st[18], scale, b, feat, dtype=torch.float64, device=dev, k_chunk=kc)
    print(f"== {name}")
    t = ws.loss_terms.double().cpu()
    for j, tn in enumerate(["depth", "color", "opacity", "feat"][:4 if feat else 3]):
        ref = o64["terms"][:, j]; sc = max(1.0, float(ref.abs().max()))
        print(f"   term {tn:8s} hip {float((t[:, j] - ref).abs().max()) / sc:9.2e}  fp32 {float((o32['terms'][:, j] - ref).abs().max()) / sc:9.2e}  (max {sc:.3g})")
    gv = arena.views(ws.grads)
    for i in range(19):
        if o64["none_grad"][i]:
            continue
        ref = o64["grads"][i]; sc = max(1e-3, float(ref.abs().max()))
        eh, eo = maxerr(gv[i], ref) / sc, maxerr(o32["grads"][i], ref) / sc
        ehh = maxerr(gv[i], o32["grads"][i]) / sc
        flag = " <<<" if eh > 1e-4 and eh > 2 * eo else (" (kink)" if eh > 1e-4 else "")
        print(f"   {ops.TENSOR_NAMES[i]:24s} hip {eh:9.2e}  fp32 {eo:9.2e}  hip-fp32 {ehh:9.2e}{flag}")
    del ws, arena, batch
    torch.cuda.empty_cache()
