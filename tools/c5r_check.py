"""Hidden-256 fused path WITHOUT the feature loss (row-split kernel A; OBJ256_FIRST_FORM=1: fwd256_kernel) against the specification
of the 16-bit modes, a few shapes, and the time of one step of 8 full-size objects.

    python tools/c5r_check.py [--full] [--time]
"""
import math
import os
import sys
import time

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
    b = synthetic.random_batch(K, R, n1, n2, seed=5 + R)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"]
    batch = {k: torch.as_tensor(b[k]).to(dev) for k in keys}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, precision=mode)
    ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize()
    first = ws.grads.clone()
    ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize()
    print(f"== {mode} K={K} R={R} S={n1 + n2}: status {int(ws.status.item())} finite {bool(torch.isfinite(ws.grads).all())} "
          f"reproducible {bool(torch.equal(first, ws.grads))}", flush=True)
    gs = 2.0 ** (math.floor(math.log2(R)) + 3) if mode == "fp16" else 1.0
    gv = arena.views(ws.grads)
    worst = 0.0
    for k in range(K):
        bk = {key: v[k:k + 1] for key, v in b.items()}
        o = oracle_step_16([p[k:k + 1] for p in st[:18]], st[18][k:k + 1], 2.0, bk, False, DT[mode], True, gs, device=dev,
                           round_head_weights=True, round_head_grads=True)
        print("  terms", np.array2string(ws.loss_terms.double().cpu().numpy()[k, :3], precision=6), "spec",
              np.array2string(o["terms"][0, :3].numpy(), precision=6))
        line = []
        for i in list(range(14)) + [18]:
            rel = rel_norm(gv[i][k], o["grads"][i][0])
            worst = max(worst, rel)
            line.append(f"{i}:{rel:.1e}")
        print("  obj", k, " ".join(line), flush=True)
        del o
        torch.cuda.empty_cache()
    return worst


def timed(dev, mode, n1=32, n2=96):
    K, R, H = 8, 8192, 256
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, H, 512, seed=3))
    b = synthetic.random_batch(K, R, n1, n2, seed=11)
    batch = {k: torch.as_tensor(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, precision=mode)
    for _ in range(2):
        ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        ops.train_step(arena, ws, batch, bf16=mode)
    torch.cuda.synchronize()
    print(f"time {mode}: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms per step of 8 objects x 8192 x {n1 + n2}", flush=True)


def main():
    dev = torch.device("cuda:0")
    shapes = [(2, 80, 16, 48), (1, 64, 8, 24), (2, 100, 32, 96), (3, 700, 8, 24), (1, 33, 16, 48)]
    if "--full" in sys.argv:
        shapes.append((2, 8192, 32, 96))
    for mode in ("bf16", "fp16"):
        for (K, R, n1, n2) in ([] if "--time-only" in sys.argv else shapes):
            w = run(dev, K, R, n1, n2, mode)
            print(f"   worst {w:.2e}")
        if "--time" in sys.argv:
            timed(dev, mode)
            if "--all-s" in sys.argv:
                timed(dev, mode, 16, 48)
                timed(dev, mode, 8, 24)


if __name__ == "__main__":
    main()
