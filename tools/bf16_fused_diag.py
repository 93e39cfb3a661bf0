#!/usr/bin/env python3
"""Where does the fused hidden-32 bf16 kernel sit between the fp32 step, the layer-wise bf16 path and the operand-rounded
specification (oracle.mlp_forward_stacked_16)?   GPU box: python tools/bf16_fused_diag.py"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import T
from openobj_amd import init as obj_init, ops, synthetic
from parity_util import oracle_step, oracle_step_16, rel_norm
dev = torch.device("cuda:0")
for K, R, n1, n2 in ((3, 300, 16, 48), (8, 2048, 16, 48)):
    st = obj_init.init_stacked(K, 32, 512, seed=11)
    arena = ops.ParamArena(K, ops.NetShape(), dev); arena.load_stacked(st)
    b = synthetic.random_batch(K, R, n1, n2, seed=5 + R)
    batch = {k: T(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
    S = n1 + n2
    res = {}
    for tag, kw, wkw in (("fused bf16", dict(bf16=True), {}), ("fused fp32", {}, {}),
                         ("layerwise bf16", dict(bf16=True, layerwise=True), dict(layerwise=True, precision="bf16"))):
        ws = ops.TrainWorkspace(arena, K, R, S, False, **wkw)
        ops.train_step(arena, ws, batch, **kw)
        torch.cuda.synchronize()
        res[tag] = [g.clone() for g in arena.views(ws.grads)]
    specs = {"spec act16": oracle_step_16(list(st[:18]), st[18], 2.0, b, False, torch.bfloat16, True, device=dev)["grads"],
             "spec op-round": oracle_step_16(list(st[:18]), st[18], 2.0, b, False, torch.bfloat16, False, device=dev)["grads"],
             "oracle fp32": oracle_step(list(st[:18]), st[18], 2.0, b, False, device=dev)["grads"]}
    print(f"== K={K} R={R} S={S}")
    print(f"   {'tensor':24s} fused-vs-spec16  fused-vs-specOp  fused-vs-layerwise16  layerwise16-vs-specOp  fused-vs-fp32  spec16-vs-fp32")
    for i in list(range(14)) + [18]:
        f = res["fused bf16"][i]
        print(f"   {ops.TENSOR_NAMES[i]:24s} {rel_norm(f, specs['spec act16'][i]):12.2e} {rel_norm(f, specs['spec op-round'][i]):14.2e}"
              f" {rel_norm(f, res['layerwise bf16'][i]):18.2e} {rel_norm(res['layerwise bf16'][i], specs['spec op-round'][i]):20.2e}"
              f" {rel_norm(f, res['fused fp32'][i]):14.2e} {rel_norm(specs['spec act16'][i], specs['oracle fp32'][i]):14.2e}")
