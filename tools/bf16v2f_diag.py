#!/usr/bin/env python3
"""Diagnostic: the second-generation bf16 kernel WITH the feature loss against (a) the fp32 fused kernel, (b) the
operand-rounded specification, (c) the first-generation kernel (OBJNERF_BF16_V1=1 in a second process: pass `save`)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from openobj_amd import ops, synthetic, init as obj_init
from parity_util import oracle_step_16, rel_norm
dev = torch.device("cuda:0")
K, R, n1, n2 = [int(x) for x in os.environ.get("SHAPE", "3,300,16,48").split(",")]
arena = ops.ParamArena(K, ops.NetShape(32, 512, 6), dev)
st = obj_init.init_stacked(K, 32, 512, seed=11)
arena.load_stacked(st)
b = synthetic.random_batch(K, R, n1, n2, seed=5 + R, feat_dim=512)
batch = {k: torch.from_numpy(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels", "gt_feat"]}
ws = ops.TrainWorkspace(arena, K, R, n1 + n2, True, precision="bf16")
ops.train_step(arena, ws, batch, with_feat=True, bf16="bf16")
torch.cuda.synchronize()
gv = arena.views(ws.grads)
tag = "v1" if os.environ.get("OBJNERF_BF16_V1") == "1" else "v2f"
path = "gpurun_out/bf16v2f_diag_%s.pt"
os.makedirs("gpurun_out", exist_ok=True)
torch.save({"grads": [g.cpu() for g in gv], "terms": ws.loss_terms.cpu()}, path % tag)
print(tag, "status", int(ws.status.item()), "finite", bool(torch.isfinite(ws.grads).all()))
print("loss terms", ws.loss_terms.cpu().numpy()[0])
if tag == "v1":
    sys.exit(0)
ws32 = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
ops.train_step(arena, ws32, batch, with_feat=True)
torch.cuda.synchronize()
g32 = arena.views(ws32.grads)
o = oracle_step_16(list(st[:18]), st[18], 2.0, b, True, torch.bfloat16, True, 1.0, device=dev)
print("fp32 terms", ws32.loss_terms.cpu().numpy()[0], "spec terms", o["terms"][0].numpy())
other = torch.load(path % "v1") if os.path.exists(path % "v1") else None
for i in range(19):
    line = f"{ops.TENSOR_NAMES[i]:24s} vs fp32 {rel_norm(gv[i], g32[i]):.3e}  vs spec {rel_norm(gv[i], o['grads'][i]):.3e}"
    if other is not None:
        line += f"  vs v1 {rel_norm(gv[i], other['grads'][i]):.3e}   (v1 vs spec {rel_norm(other['grads'][i], o['grads'][i]):.3e})"
    print(line)
if other is not None:
    print("terms v1", other["terms"].numpy()[0])
