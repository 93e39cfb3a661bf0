#!/usr/bin/env python3
"""Fixture G9D: the REFERENCE's PSNR after 10 and 20 iterations on scene G9C (hidden 256, BASELINE configs[4]'s network),
one row per weight seed of G9C.  At this width training is chaotic by iteration 50 (profiles/r04_h256_sensitivity.txt:
a 1e-7 perturbation of the initial weights moves the reference's own PSNR by 0.3 .. 2.7 dB), but NOT yet after 10
iterations: the same perturbation moves it by <= 0.003 dB there and by <= 0.05 dB after 20
(tools/h256_early.py -> profiles/r05_h256_early_sensitivity.txt), while the PSNR has already risen from ~10 to ~20 dB.  A
PAIRED, per-seed comparison is therefore well-posed at these counts and resolves far below the 0.1 dB the metric names
(tests/test_psnr_gpu.py::test_hidden_256_network_psnr_paired_early).

    THREADS=4 python tests/golden/make_g9d_early.py [n_seeds]      -> tests/golden/g9d_early_h256.npz

Build container only (imports /root/reference through make_golden.py)."""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_g9b_ensemble as G  # noqa: E402
import make_golden as MG  # noqa: E402

ITERS = (10, 20)


def main(n_seeds):
    torch.set_num_threads(int(os.environ.get("THREADS", "4")))
    spec = G.G9C
    scene = MG.synthetic.EllipsoidScene.make(spec["K"], 512, seed=spec["scene_seed"])
    ev = scene.eval_rays(spec["eval_R"], spec["eval_S"])
    cache = {}

    def batches(it):
        if it not in cache:
            cache[it] = scene.batch(spec["R"], spec["N"], spec["M"], seed=spec["batch_seed"] + it, with_feat=True)
        return cache[it]

    rows = []
    for seed in range(spec["weight_seed"], spec["weight_seed"] + n_seeds):
        ts = MG.make_trainers(spec["K"], seed=seed, perturb_B=False, hidden=spec["hidden"])
        out = {}

        def on_step(done, fc_param, pe_param):
            if done in ITERS:
                out[done] = MG._g9_eval(ts, [p.detach().clone() for p in fc_param], pe_param[0].detach().clone(), ev)[0]

        MG.run_reference_steps(ts, batches, False, n_steps=max(ITERS), record_grads=False, on_step=on_step)
        rows.append((seed, out[10], out[20]))
        print("g9d", *rows[-1], flush=True)
    r = np.array(rows, np.float64)
    np.savez(os.path.join(HERE, "g9d_early_h256.npz"), seeds=r[:, 0].astype(np.int64), psnr10=r[:, 1], psnr20=r[:, 2],
             meta=np.array([spec["K"], spec["R"], spec["N"], spec["M"], spec["hidden"], spec["eval_R"], spec["eval_S"]]))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 33)
