#!/usr/bin/env python3
"""Reference PSNR ensemble of the G9 scene (tests/golden/make_golden.py g9): the reference's own 300-iteration
training run for N weight seeds, PSNR on the held-out rays.  Training is chaotic (a 1e-7 perturbation moves a single
run by ~0.5 dB), so the "PSNR within 0.1 dB of the reference" claim is a statement about ENSEMBLE MEANS; this fixture
is the reference side of it.      python tests/golden/make_g9_ensemble.py [first_seed n_seeds out_name]
Build container only (imports /root/reference)."""
import os, sys
import numpy as np
import torch
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG

first = int(sys.argv[1]) if len(sys.argv) > 1 else MG.G9["weight_seed"]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 128
name = sys.argv[3] if len(sys.argv) > 3 else "g9_ensemble"
torch.set_num_threads(int(os.environ.get("THREADS", "2")))
scene = MG.synthetic.EllipsoidScene.make(MG.G9["K"], 512, seed=MG.G9["scene_seed"])
ev = scene.eval_rays(MG.G9["eval_R"], MG.G9["eval_S"])
batches = lambda it: scene.batch(MG.G9["R"], MG.G9["N"], MG.G9["M"], seed=9000 + it, with_feat=True)
out = []
for seed in range(first, first + n):
    ts = MG.make_trainers(MG.G9["K"], seed=seed, perturb_B=False)
    rec = MG.run_reference_steps(ts, batches, False, n_steps=MG.G9["steps"], record_grads=False)
    out.append(MG._g9_eval(ts, rec["final_fc"], rec["final_B"], ev)[0])
    print(seed, out[-1], flush=True)
np.savez(os.path.join(HERE, name + ".npz"), seeds=np.arange(first, first + n), psnr=np.array(out))
