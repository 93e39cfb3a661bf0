#!/usr/bin/env python3
"""Build container only (imports the reference): how far does the REFERENCE's own hidden-256 PSNR after 50 iterations
(scene G9C) move under a 1e-7 relative perturbation of its initial weights?  Shows that a per-seed comparison is not
well-posed at this width, unlike hidden 32 (tests/test_psnr_gpu.py).  Output: profiles/r04_h256_sensitivity.txt"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests", "golden"))
import make_g9b_ensemble as G
import make_golden as MG
torch.set_num_threads(int(os.environ.get("THREADS", "4")))
spec = G.G9C
scene = MG.synthetic.EllipsoidScene.make(spec["K"], 512, seed=spec["scene_seed"])
ev = scene.eval_rays(spec["eval_R"], spec["eval_S"])
cache = {}
def batches(it):
    if it not in cache:
        cache[it] = scene.batch(spec["R"], spec["N"], spec["M"], seed=spec["batch_seed"] + it, with_feat=True)
    return cache[it]
ref = np.load(os.path.join(HERE, "..", "tests", "golden", "g9c_ensemble_h256.npz"))
for seed in [int(s) for s in sys.argv[1:]] or [9000, 9001, 9002]:
    ts = MG.make_trainers(spec["K"], seed=seed, perturb_B=False, hidden=spec["hidden"])
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for t in ts:
            for p in t.fc_occ_map.parameters():
                p.mul_(1.0 + 1e-7 * torch.randn(p.shape, generator=g))
    rec = MG.run_reference_steps(ts, batches, False, n_steps=spec["early"], record_grads=False)
    p50 = MG._g9_eval(ts, rec["final_fc"], rec["final_B"], ev)[0]
    base = float(ref["psnr50"][list(ref["seeds"]).index(seed)])
    print(f"seed {seed}: reference PSNR after 50 iterations {base:.4f} dB; same seed, initial weights x (1 + 1e-7 N(0,1)): {p50:.4f} dB; delta {p50 - base:+.4f} dB", flush=True)
