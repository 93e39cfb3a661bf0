"""PSNR distribution of the integration fixture over weight seeds, fp32 vs bf16 mode (GPU)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from openobj_amd import synthetic
from test_api_gpu import _train_and_psnr

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "g9_psnr_nofeat.npz"))
K, R, N, M, steps, eval_R, eval_S, scene_seed = [int(x) for x in g["meta"]]
dev = torch.device("cuda:0")
scene = synthetic.EllipsoidScene.make(K, 512, seed=scene_seed)
ev = scene.eval_rays(eval_R, eval_S)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for bf16 in (False, True):
    ens = np.array([_train_and_psnr(dev, scene, (K, R, N, M), 90 + i, steps, ev, bf16=bf16)[0] for i in range(n)])
    print("bf16" if bf16 else "fp32", "mean %.3f median %.3f std %.3f min %.3f max %.3f" %
          (ens.mean(), np.median(ens), ens.std(), ens.min(), ens.max()))
    print(np.round(ens, 2))
