#!/usr/bin/env python3
"""Build container only (imports the reference): how far does the REFERENCE's own PSNR after 50 iterations move under a
1e-7 relative perturbation of its initial weights?
  h256_sensitivity.py seeds...          scene G9C (hidden 256): shows that a per-seed comparison is not well-posed at that
                                        width, unlike hidden 32 (tests/test_psnr_gpu.py) -> profiles/r04_h256_sensitivity.txt
  VARIANT=feat h256_sensitivity.py ...  scene G9B with the 512-d feature loss (hidden 32): the one seed of 128 whose HIP run
                                        differs by 0.136 dB (seed 9092) -> profiles/r04_feat_seed9092_sensitivity.txt"""
import os, sys
import numpy as np, torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "tests", "golden"))
import make_g9b_ensemble as G
import make_golden as MG
torch.set_num_threads(int(os.environ.get("THREADS", "4")))
VARIANT = os.environ.get("VARIANT", "h256")
spec = G.G9C if VARIANT == "h256" else G.G9B
FEAT = VARIANT == "feat"
PERT = [int(x) for x in os.environ.get("PERT_SEEDS", "1").split(",")]
EPS = float(os.environ.get("EPS", "1e-7"))
scene = MG.synthetic.EllipsoidScene.make(spec["K"], 512, seed=spec["scene_seed"])
ev = scene.eval_rays(spec["eval_R"], spec["eval_S"])
cache = {}
def batches(it):
    if it not in cache:
        cache[it] = scene.batch(spec["R"], spec["N"], spec["M"], seed=spec["batch_seed"] + it, with_feat=True)
    return cache[it]
ref = np.load(os.path.join(HERE, "..", "tests", "golden", "g9c_ensemble_h256.npz" if VARIANT == "h256" else
                           "g9b_ensemble_%s.npz" % VARIANT))
for seed in [int(s) for s in sys.argv[1:]] or [9000, 9001, 9002]:
  for ps in PERT:
    ts = MG.make_trainers(spec["K"], seed=seed, perturb_B=False, hidden=spec["hidden"])
    g = torch.Generator().manual_seed(ps)
    with torch.no_grad():
        for t in ts:
            for p in t.fc_occ_map.parameters():
                p.mul_(1.0 + EPS * torch.randn(p.shape, generator=g))
    rec = MG.run_reference_steps(ts, batches, FEAT, n_steps=spec["early"], record_grads=False)
    p50 = MG._g9_eval(ts, rec["final_fc"], rec["final_B"], ev)[0]
    base = float(ref["psnr50"][list(ref["seeds"]).index(seed)])
    print(f"seed {seed}: reference PSNR after 50 iterations {base:.4f} dB; same seed, initial weights x (1 + {EPS:g} N(0,1)) [draw {ps}]: {p50:.4f} dB; delta {p50 - base:+.4f} dB", flush=True)
