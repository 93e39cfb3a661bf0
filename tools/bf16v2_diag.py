#!/usr/bin/env python3
"""Diagnostic: per-tensor / per-column distance of the fused bf16 step from its operand-rounded specification."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from openobj_amd import ops, synthetic, init as obj_init
from parity_util import oracle_step_16, rel_norm
dev = torch.device("cuda:0")
K, R, n1, n2 = 3, 300, 16, 48
arena = ops.ParamArena(K, ops.NetShape(32, 512, 6), dev)
st = obj_init.init_stacked(K, 32, 512, seed=11)
arena.load_stacked(st)
b = synthetic.random_batch(K, R, n1, n2, seed=5 + R)
batch = {k: torch.from_numpy(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, precision="bf16")
ops.train_step(arena, ws, batch, bf16="bf16")
torch.cuda.synchronize()
o = oracle_step_16(list(st[:18]), st[18], 2.0, b, False, torch.bfloat16, True, 1.0, device=dev, round_head_grads=True)
gv = arena.views(ws.grads)
print("loss terms", ws.loss_terms.cpu().numpy()[0], o["terms"][0].numpy())
for i in list(range(14)) + [18]:
    print(f"{ops.TENSOR_NAMES[i]:24s} rel {rel_norm(gv[i], o['grads'][i]):.3e}")
for i in (10, 4, 0):
    a, r = gv[i].cpu().double(), o["grads"][i].double()
    print(ops.TENSOR_NAMES[i], "per column rel:")
    e = ((a - r).norm(dim=(0, 1)) / (r.norm(dim=(0, 1)) + 1e-30)).numpy()
    print(np.array2string(e, precision=2, max_line_width=200))
print("cl bias ours  ", gv[11][0].cpu().numpy()[:8])
print("cl bias oracle", o["grads"][11][0].numpy()[:8])
print("in bias ours  ", gv[1][0].cpu().numpy()[:8])
print("in bias oracle", o["grads"][1][0].numpy()[:8])
