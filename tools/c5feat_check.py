"""Fused hidden-256 path WITH the feature loss (objnerf_train256.hip, FEAT instantiations) against the specification of
the 16-bit modes (tests/parity_util.oracle_step_16): per-tensor relative errors and loss terms, a few shapes.

    python tools/c5feat_check.py [--full]        (GPU; --full adds 8192 rays x 128 samples)
"""
import math
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import init as obj_init  # noqa: E402
from openobj_amd import ops, synthetic  # noqa: E402
from parity_util import oracle_step_16, rel_norm  # noqa: E402

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}


def run(dev, K, R, n1, n2, mode, seed=7):
    H = 256
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    st = obj_init.init_stacked(K, H, 512, seed=seed)
    arena.load_stacked(st)
    b = synthetic.random_batch(K, R, n1, n2, seed=5 + R, feat_dim=512)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels", "gt_feat"]
    batch = {k: torch.as_tensor(b[k]).to(dev) for k in keys}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, True, precision=mode)
    ops.train_step(arena, ws, batch, with_feat=True, bf16=mode)
    torch.cuda.synchronize()
    first = ws.grads.clone()
    ops.train_step(arena, ws, batch, with_feat=True, bf16=mode)
    torch.cuda.synchronize()
    print(f"== {mode} K={K} R={R} S={n1 + n2}: status {int(ws.status.item())} finite {bool(torch.isfinite(ws.grads).all())} "
          f"reproducible {bool(torch.equal(first, ws.grads))}")
    gs = 2.0 ** (math.floor(math.log2(R)) + 3) if mode == "fp16" else 1.0
    gv = arena.views(ws.grads)
    worst = 0.0
    for k in range(K):
        bk = {key: v[k:k + 1] for key, v in b.items()}
        o = oracle_step_16([p[k:k + 1] for p in st[:18]], st[18][k:k + 1], 2.0, bk, True, DT[mode], True, gs, device=dev,
                           round_head_weights=True, round_head_grads=True)
        print("  terms", np.array2string(ws.loss_terms.double().cpu().numpy()[k], precision=6), "spec",
              np.array2string(o["terms"][0].numpy(), precision=6))
        line = []
        for i in range(19):
            rel = rel_norm(gv[i][k], o["grads"][i][0])
            worst = max(worst, rel)
            line.append(f"{i}:{rel:.1e}")
        print("  obj", k, " ".join(line))
        del o
        torch.cuda.empty_cache()
    return worst


def main():
    dev = torch.device("cuda:0")
    shapes = [(2, 80, 16, 48), (1, 64, 8, 24), (2, 100, 32, 96), (3, 700, 8, 24)]
    if "--full" in sys.argv:
        shapes.append((2, 8192, 32, 96))
    for mode in ("bf16", "fp16"):
        for (K, R, n1, n2) in shapes:
            w = run(dev, K, R, n1, n2, mode)
            print(f"   worst {w:.2e}")


if __name__ == "__main__":
    main()
