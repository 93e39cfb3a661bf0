#!/usr/bin/env python3
"""Build container only (imports the reference).  Scene G9C (hidden 256): the REFERENCE's PSNR after a FEW iterations, per
seed, with and without a 1e-7 relative perturbation of its initial weights -- finds the iteration count at which a PAIRED
(per-seed) comparison of two fp32 implementations is still well-posed at this width (at 50 iterations it is not:
profiles/r04_h256_sensitivity.txt).
    THREADS=4 ITERS=5,10,20 h256_early.py seed...     -> one line per (seed, perturbation draw, iteration count)"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests", "golden"))
import make_g9b_ensemble as G
import make_golden as MG
torch.set_num_threads(int(os.environ.get("THREADS", "4")))
ITERS = [int(x) for x in os.environ.get("ITERS", "5,10,20").split(",")]
EPS = float(os.environ.get("EPS", "1e-7"))
spec = G.G9C
scene = MG.synthetic.EllipsoidScene.make(spec["K"], 512, seed=spec["scene_seed"])
ev = scene.eval_rays(spec["eval_R"], spec["eval_S"])
cache = {}
def batches(it):
    if it not in cache:
        cache[it] = scene.batch(spec["R"], spec["N"], spec["M"], seed=spec["batch_seed"] + it, with_feat=True)
    return cache[it]
for seed in [int(s) for s in sys.argv[1:]] or [9000, 9001]:
    res = {}
    for ps in (0, 1):
        ts = MG.make_trainers(spec["K"], seed=seed, perturb_B=False, hidden=spec["hidden"])
        if ps:
            g = torch.Generator().manual_seed(ps)
            with torch.no_grad():
                for t in ts:
                    for p in t.fc_occ_map.parameters():
                        p.mul_(1.0 + EPS * torch.randn(p.shape, generator=g))
        out = {}
        def on_step(done, fc_param, pe_param):
            if done in ITERS:
                out[done] = MG._g9_eval(ts, [p.detach().clone() for p in fc_param], pe_param[0].detach().clone(), ev)[0]
        MG.run_reference_steps(ts, batches, False, n_steps=max(ITERS), record_grads=False, on_step=on_step)
        res[ps] = out
    for it in ITERS:
        print(f"seed {seed} iter {it}: reference {res[0][it]:.4f} dB; initial weights x (1 + {EPS:g} N(0,1)): {res[1][it]:.4f} dB; delta {res[1][it] - res[0][it]:+.4f} dB", flush=True)
