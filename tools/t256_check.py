#!/usr/bin/env python3
"""Fused hidden-256 16-bit path (objnerf_train256.hip) against its specification and against the layer-wise chain.
GPU box:  python tools/t256_check.py [K R n1 n2 mode]"""
import math, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import T
from openobj_amd import init as obj_init, ops, synthetic
from parity_util import oracle_step_16, rel_norm
dev = torch.device("cuda:0")
args = sys.argv[1:]
K, R, n1, n2 = [int(x) for x in args[:4]] if len(args) >= 4 else (1, 64, 16, 48)
mode = args[4] if len(args) > 4 else "bf16"
H, S = 256, n1 + n2
st = obj_init.init_stacked(K, H, 512, seed=7)
arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev); arena.load_stacked(st)
b = synthetic.random_batch(K, R, n1, n2, seed=5 + R)
batch = {k: T(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
ws = ops.TrainWorkspace(arena, K, R, S, False, precision=mode)
ops.train_step(arena, ws, batch, bf16=mode)
torch.cuda.synchronize()
print("status", int(ws.status.item()), "finite", bool(torch.isfinite(ws.grads).all()))
print("fused done", flush=True)
wl = ops.TrainWorkspace(arena, K, R, S, False, precision=mode, layerwise=True)
ops.train_step(arena, wl, batch, bf16=mode, layerwise=True)
torch.cuda.synchronize()
print("layerwise done", flush=True)
gs = 2.0 ** (math.floor(math.log2(R)) + 3) if mode == "fp16" else 1.0
dt = torch.bfloat16 if mode == "bf16" else torch.float16
o = oracle_step_16(list(st[:18]), st[18], 2.0, b, False, dt, True, gs, device=dev, round_head_weights=True, round_head_grads=True)
print("loss terms fused   ", ws.loss_terms.cpu().tolist())
print("loss terms layerw. ", wl.loss_terms.cpu().tolist())
print("loss terms spec    ", o["terms"].tolist())
gv, gl = arena.views(ws.grads), arena.views(wl.grads)
for i in list(range(14)) + [18]:
    print(f"{ops.TENSOR_NAMES[i]:24s} fused-vs-spec {rel_norm(gv[i], o['grads'][i]):9.2e}   fused-vs-layerwise {rel_norm(gv[i], gl[i]):9.2e}   |spec| {float(o['grads'][i].norm()):9.2e} |fused| {float(gv[i].norm()):9.2e}")
