"""Diagnostic (GPU): one weight seed of the G9b feature-loss scene trained side by side by the HIP fp32 step and by the
oracle (torch fp32 AND fp64 on the device) from the same initial weights and batches: per iteration the loss difference and
the largest parameter difference, per tensor at the first iteration where they separate.  Usage: psnr_seed_trace.py SEED"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from openobj_amd import ops, optim, psnr_scene
from oracle import objnerf_oracle as O
dev = torch.device("cuda:0")
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 9092
er = psnr_scene.EnsembleRun(dev, with_feat=True)
s = er.spec
arena = er.initial_arena([seed])
K = arena.K
ws = ops.TrainWorkspace(arena, K, s["R"], s["N"] + s["M"], True)
opt = optim.ArenaAdamW(arena, lr=er.cfg.learning_rate, weight_decay=er.cfg.weight_decay)
mask = arena.has_grad_mask(True)

def oracle_state(dt):
    p = [v.clone().to(dt) for v in arena.views()]
    return dict(p=p, m=[torch.zeros_like(x) for x in p], v=[torch.zeros_like(x) for x in p])
st32, st64 = oracle_state(torch.float32), oracle_state(torch.float64)
scale = arena.scale.clone()

def oracle_step(st, b, it, dt):
    fc = [x.clone().requires_grad_(True) for x in st["p"][:18]]
    B = st["p"][18].clone().requires_grad_(True)
    if dt == torch.float64:
        loss, _ = O.train_forward_loss([x.float() for x in fc], B.float(), scale, b["pts"], b["gt_depth"], b["gt_rgb"], b["labels"], b["z"],
                                       gt_feat=b["gt_feat"], mlp_dtype=torch.float64) if False else (None, None)
    loss, _ = O.train_forward_loss(fc if dt == torch.float32 else fc, B, scale if dt == torch.float32 else scale.double(),
                                   b["pts"] if dt == torch.float32 else b["pts"].double(), b["gt_depth"].to(dt), b["gt_rgb"].to(dt), b["labels"],
                                   b["z"].to(dt), gt_feat=b["gt_feat"].to(dt))
    grads = torch.autograd.grad(loss, fc + [B], allow_unused=True)
    with torch.no_grad():
        for p, g, m, v in zip(st["p"], grads, st["m"], st["v"]):
            if g is not None:
                O.adamw_step(p, g, m, v, it + 1, er.cfg.learning_rate, er.cfg.weight_decay)
    return float(loss)

for it, b in enumerate(er.batches[:s["early"]]):
    ops.train_step(arena, ws, b, with_feat=True)
    t = ws.loss_terms.double().cpu()
    lh = float((t[:, 0] + 5 * t[:, 1] + 10 * t[:, 2] + 5 * t[:, 3]).sum())
    opt.step(ws.grads, mask, flags=ws.flags)
    l32 = oracle_step(st32, b, it, torch.float32)
    l64 = oracle_step(st64, b, it, torch.float64)
    hv = arena.views()
    d_h64 = max(float((hv[i].double() - st64["p"][i]).abs().max()) for i in range(19))
    d_3264 = max(float((st32["p"][i].double() - st64["p"][i]).abs().max()) for i in range(19))
    worst = max(range(19), key=lambda i: float((hv[i].double() - st64["p"][i]).abs().max()))
    print("it %2d  loss hip %.6f o32 %.6f o64 %.6f   max|p_hip - p_64| %.2e (%s)   max|p_o32 - p_64| %.2e" %
          (it, lh, l32, l64, d_h64, ops.TENSOR_NAMES[worst], d_3264), flush=True)
ev_h = er._evaluate(arena, 1)["psnr"][0]
def psnr_of(st):
    a2 = ops.ParamArena(K, arena.net, dev)
    a2.load_stacked([x.float() for x in st["p"]]); a2.scale.copy_(arena.scale)
    return er._evaluate(a2, 1)["psnr"][0]
print("PSNR50  hip %.4f  oracle fp32 %.4f  oracle fp64 %.4f" % (ev_h, psnr_of(st32), psnr_of(st64)))
