"""The PSNR scene of BASELINE.json's metric ("... PSNR delta vs ref"; SURVEY.md 8(d), fixture G9): K analytic
ellipsoids (synthetic.EllipsoidScene), the reference's initial weights for a seed, the seeded batches of
tests/golden/make_golden.py g9, `steps` fused iterations, PSNR of the rendered colour on held-out rays.

The reference computes no PSNR anywhere; the reference side of the comparison is the ensemble its own modules
produced for the same seeds (tests/golden/g9_ensemble.npz, written by tests/golden/make_g9_ensemble.py).  Training is
chaotic -- a 1e-7 relative perturbation of the initial weights moves the reference's own 300-iteration PSNR by
~0.5 dB -- so the delta is a difference of ENSEMBLE MEANS with a confidence interval, never a single run.
"""
import math
import os
from typing import Dict, List, Optional

import numpy as np
import torch

from . import cfg as ocfg
from . import ops, synthetic, trainer
from . import train as otrain

G9 = dict(K=4, R=96, N=4, M=12, steps=300, eval_R=256, eval_S=32, scene_seed=7, weight_seed=90)
ENSEMBLE_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden",
                                "g9_ensemble.npz")


def _psnr(pred: torch.Tensor, gt: torch.Tensor) -> float:
    mse = torch.mean((pred - gt) ** 2).item()
    return -10.0 * math.log10(max(mse, 1e-20))


class PsnrScene:
    def __init__(self, device, steps: Optional[int] = None):
        self.dev = torch.device(device)
        self.steps = steps or G9["steps"]
        self.scene = synthetic.EllipsoidScene.make(G9["K"], 512, seed=G9["scene_seed"])
        ev = self.scene.eval_rays(G9["eval_R"], G9["eval_S"])
        self.ev = {k: torch.from_numpy(ev[k]).to(self.dev) for k in ("pts", "z", "gt_rgb")}
        keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"]
        self.batches = []
        for it in range(self.steps):                       # the same seeded batches for every weight seed
            b = self.scene.batch(G9["R"], G9["N"], G9["M"], seed=9000 + it)
            self.batches.append({k: torch.from_numpy(b[k]).to(self.dev) for k in keys})
        c = ocfg.Config(ocfg.replica_room0_config(train_device=str(self.dev)))
        c.obj_id = 1
        self.cfg = c

    def run(self, seed: int, bf16=False) -> float:
        """Train K fresh object networks (the reference's initialisation for `seed`) and return the PSNR [dB].
        bf16: operand precision (ops.precision_bits: False, True / "bf16", "fp16")."""
        torch.manual_seed(seed)
        ts = [trainer.Trainer(self.cfg) for _ in range(G9["K"])]
        loop = otrain.HipTrainLoop(self.cfg, ts, with_feat=False, bf16=bf16)
        for b in self.batches:
            loop.step(b)
        R, S = G9["eval_R"], G9["eval_S"]
        a, c, _, _ = ops.eval_points(loop.arena, self.ev["pts"].reshape(G9["K"], -1, 3))
        out = ops.composite(a.reshape(-1, S), c.reshape(-1, S, 3), self.ev["z"].reshape(-1, S))
        return _psnr(out["rgb"].reshape(G9["K"], R, 3), self.ev["gt_rgb"])

    def ensemble(self, seeds: List[int], bf16=False) -> np.ndarray:
        return np.array([self.run(s, bf16) for s in seeds])


def reference_ensemble() -> Optional[Dict[str, np.ndarray]]:
    """seeds / psnr arrays of the reference's own runs (None when the fixture is absent)."""
    try:
        d = np.load(ENSEMBLE_FIXTURE)
        return {"seeds": d["seeds"], "psnr": d["psnr"]}
    except OSError:
        return None


def delta_report(hip: np.ndarray, ref: np.ndarray) -> Dict[str, float]:
    """Difference of ensemble means with its 95 % confidence half-width (Welch, normal quantile)."""
    se = math.sqrt(hip.var(ddof=1) / len(hip) + ref.var(ddof=1) / len(ref))
    return {"delta_db": float(hip.mean() - ref.mean()), "ci95_db": 1.96 * se, "hip_mean_db": float(hip.mean()),
            "ref_mean_db": float(ref.mean()), "hip_std_db": float(hip.std(ddof=1)), "ref_std_db": float(ref.std(ddof=1)),
            "n_hip": int(len(hip)), "n_ref": int(len(ref))}
