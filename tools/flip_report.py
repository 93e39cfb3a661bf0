#!/usr/bin/env python3
"""ReLU branch flips against the fp64 anchor: the fused fp32 kernel, the layer-wise HIP path and torch's own fp32
arithmetic, per layer, next to the number of units whose anchor pre-activation lies within each band of zero
(tests/parity_util.py BANDS).  GPU box:   python tools/flip_report.py  > profiles/r03_flip_report.txt
An implementation whose pre-activation error is <= e everywhere can flip at most near(e) units; the fused kernel's
angle-doubled embedding octaves carry ~2e-6 of error, the layer-wise path and torch ~3e-7."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import T
from openobj_amd import init as obj_init, ops, synthetic
from oracle import objnerf_oracle as O
from parity_util import BANDS, oracle_step, unpack_masks

dev = torch.device("cuda:0")
CONFIGS = [("headline shape K=10 R=4096 S=64", 10, 4096, 16, 48, 32, False, 5),
           ("c4 share K=15 R=4096 S=64 feat", 15, 4096, 16, 48, 32, True, 3),
           ("native K=50 R=120 S=10", 50, 120, 1, 9, 32, False, 50)]
LAYERS = ["h1", "h2", "h3", "h4", "hc", "hf"]
for name, K, R, n1, n2, H, feat, kc in CONFIGS:
    S = n1 + n2
    st = obj_init.init_stacked(K, H, 512, seed=123)
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    arena.load_stacked(st)
    b = synthetic.random_batch(K, R, n1, n2, seed=321, feat_dim=512 if feat else 0)
    b["labels"][:, 0] = 1
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    batch = {k: T(b[k]).to(dev) for k in keys}
    rows = {}
    for tag, lw in (("fused", False), ("layer-wise", True)):
        ws = ops.TrainWorkspace(arena, K, R, S, feat, layerwise=lw)
        mb = torch.zeros(K, R, S, 6, H // 8, dtype=torch.uint8, device=dev)
        ops.train_step(arena, ws, batch, with_feat=feat, layerwise=lw, relu_masks=mb)
        torch.cuda.synchronize()
        o = oracle_step(list(st[:18]), st[18], 2.0, b, feat, dtype=torch.float64, device=dev, k_chunk=kc,
                        masks=unpack_masks(mb, H))
        rows[tag] = o
        del ws, mb
    # torch's own fp32 arithmetic against the same anchor
    nl = 6 if feat else 5
    tf = [0] * nl
    for k0 in range(0, K, kc):
        sl = slice(k0, min(K, k0 + kc))
        fc = [p[sl].to(dev) for p in st[:18]]
        emb = O.embed_stacked(st[18][sl].to(dev), torch.full((sl.stop - sl.start,), 2.0, device=dev), batch["pts"][sl])
        with torch.no_grad():
            p32 = O.mlp_forward_stacked(fc, emb, feat, None, True)[3]
            p64 = O.mlp_forward_stacked([p.double() for p in fc], emb.double(), feat, None, True)[3]
        for l in range(nl):
            tf[l] += int(((p32[l] > 0) != (p64[l] > 0)).sum())
        del p32, p64
    print(f"== {name}   units per layer {K * R * S * H:.3g}")
    print("   layer   fused  layer-wise  torch-fp32 |  units within " + "  ".join(f"{w:.0e}" for w in BANDS) + "  | max |x| of a fused flip")
    for l in range(nl):
        nr = rows["fused"]["near"][l]
        print(f"   {LAYERS[l]:5s} {rows['fused']['flips'][l][0]:7d} {rows['layer-wise']['flips'][l][0]:11d} {tf[l]:11d} |"
              f"               " + "  ".join(f"{v:5d}" for v in nr) + f"  | {rows['fused']['flips'][l][1]:.1e}")
    torch.cuda.empty_cache()
