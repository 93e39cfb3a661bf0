"""Diagnostic: per-seed PSNR differences (fp32, with the feature loss, 50 iterations) against the reference fixture; prints the
seeds beyond 0.05 dB."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openobj_amd import psnr_scene
dev = torch.device("cuda:0")
ref = psnr_scene.reference_ensemble_b(True)
seeds = [int(x) for x in ref["seeds"]]
run = psnr_scene.EnsembleRun(dev, with_feat=True).run(seeds, False)
d = run["psnr50"] - ref["psnr50"][:len(seeds)]
for s, x in zip(seeds, d):
    if abs(x) > 0.05:
        print("seed", s, "delta %.4f dB" % x)
print("std %.4f max %.4f" % (d.std(ddof=1), np.abs(d).max()))
