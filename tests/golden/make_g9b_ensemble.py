#!/usr/bin/env python3
"""Reference PSNR ensembles of the scene SURVEY.md 8(d) specifies ("G9b"): K = 8 analytic ellipsoids with constant
colour and a constant unit feature, the reference's own modules trained for 300 iterations on seeded batches of
96 rays x 16 samples per object, PSNR of the rendered colour on 4096 held-out label-1 rays per object -- recorded after
50 iterations (trajectories of two correct fp32 implementations have not diverged yet: a per-seed comparison is
well-posed) AND after 300 (chaotic: only ensemble means compare), for every weight seed; with and without the 512-d
feature loss (cfg.part_mode).  The feature variant also records the mean cosine between the rendered 512-d feature
(the reference's composited out_clip tensor) and the target on the first 256 held-out rays of each object.

    python tests/golden/make_g9b_ensemble.py run  nofeat|feat|h256 first_seed n_seeds part_file
    python tests/golden/make_g9b_ensemble.py join nofeat|feat|h256 part_file...   -> g9b_ensemble_<variant>.npz
                                                                                     (h256: g9c_ensemble_h256.npz)
    python tests/golden/make_g9b_ensemble.py all  [n_nofeat n_feat n_procs]       (shards over processes, then joins)

Build container only (imports /root/reference through make_golden.py)."""
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

G9B = dict(K=8, R=96, N=4, M=12, steps=300, early=50, eval_R=4096, eval_S=32, feat_R=256, scene_seed=7,
           weight_seed=9000, batch_seed=9000, hidden=32)
# "h256": the same scene for BASELINE configs[4]'s network (hidden 256) at 32 samples per ray, no feature loss -- the
# shape the fused hidden-256 kernels of the 16-bit modes take (objnerf_train256.hip); fewer seeds (the reference costs
# ~60 GFLOP per iteration on the CPU)
G9C = dict(G9B, N=8, M=24, hidden=256)


def run(variant, first, n, out):
    import make_golden as MG
    torch.set_num_threads(int(os.environ.get("THREADS", "2")))
    feat_on = variant == "feat"
    G9B = G9C if variant == "h256" else globals()["G9B"]
    scene = MG.synthetic.EllipsoidScene.make(G9B["K"], 512, seed=G9B["scene_seed"])
    ev = scene.eval_rays(G9B["eval_R"], G9B["eval_S"])
    cache = {}

    def batches(it):
        if it not in cache:
            cache[it] = scene.batch(G9B["R"], G9B["N"], G9B["M"], seed=G9B["batch_seed"] + it, with_feat=True)
        return cache[it]

    def feat_cos(ts):
        """Mean cosine of the reference's rendered feature (render of the [R,S,512] out_clip tensor, loss.py:82) with
        the target, first feat_R held-out rays per object."""
        pts, = MG.to_t({"pts": ev["pts"][:, :G9B["feat_R"]]}, ["pts"])
        cs = []
        with torch.no_grad():
            for k, t in enumerate(ts):
                a, _, f = t.fc_occ_map(t.pe(pts[k]))
                term = MG.ref_rr.occupancy_to_termination(MG.ref_rr.occupancy_activation(a.squeeze(-1)))
                F = MG.ref_rr.render(term[..., None], f, dim=-2)
                g = torch.from_numpy(scene.feat[k])[None]
                cs.append(torch.nn.functional.cosine_similarity(F, g, dim=-1).mean().item())
        return float(np.mean(cs))

    rows = []
    for seed in range(first, first + n):
        ts = MG.make_trainers(G9B["K"], seed=seed, perturb_B=False, hidden=G9B["hidden"])
        early = {}

        def on_step(done, fc_param, pe_param):
            if done == G9B["early"]:
                early["psnr"] = MG._g9_eval(ts, [p.detach().clone() for p in fc_param], pe_param[0].detach().clone(), ev)[0]

        rec = MG.run_reference_steps(ts, batches, feat_on, n_steps=G9B["steps"], record_grads=False, on_step=on_step)
        p300 = MG._g9_eval(ts, rec["final_fc"], rec["final_B"], ev)[0]      # (copies the final weights into ts)
        fc = feat_cos(ts) if feat_on else 0.0
        rows.append((seed, early["psnr"], p300, fc, rec["loss"][-1]))
        print(variant, *rows[-1], flush=True)
    np.save(out, np.array(rows, np.float64))


def join(variant, parts):
    rows = np.concatenate([np.load(p) for p in parts])
    rows = rows[np.argsort(rows[:, 0])]
    G9B = G9C if variant == "h256" else globals()["G9B"]
    np.savez(os.path.join(HERE, ("g9c_ensemble_h256.npz" if variant == "h256" else f"g9b_ensemble_{variant}.npz")), seeds=rows[:, 0].astype(np.int64), psnr50=rows[:, 1],
             psnr300=rows[:, 2], featcos300=rows[:, 3], loss300=rows[:, 4],
             meta=np.array([G9B[k] for k in ["K", "R", "N", "M", "steps", "early", "eval_R", "eval_S", "feat_R",
                                             "scene_seed", "batch_seed", "hidden"]], np.int64))
    print(variant, len(rows), "seeds  PSNR50 mean", rows[:, 1].mean(), " PSNR300 mean", rows[:, 2].mean(), "sigma",
          rows[:, 2].std(ddof=1))


def all_(n_nofeat=320, n_feat=128, n_procs=4):
    import tempfile
    tmp = tempfile.mkdtemp(prefix="g9b_")
    for variant, n in (("nofeat", n_nofeat), ("feat", n_feat)):
        per = (n + n_procs - 1) // n_procs
        jobs, parts = [], []
        for i in range(n_procs):
            lo = G9B["weight_seed"] + i * per
            cnt = min(per, G9B["weight_seed"] + n - lo)
            if cnt <= 0:
                continue
            part = os.path.join(tmp, f"{variant}_{i}.npy")
            parts.append(part)
            jobs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "run", variant, str(lo), str(cnt), part]))
        assert all(j.wait() == 0 for j in jobs)
        join(variant, parts)


if __name__ == "__main__":
    cmd = sys.argv[1] if len(sys.argv) > 1 else "all"
    if cmd == "run":
        run(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5])
    elif cmd == "join":
        join(sys.argv[2], sys.argv[3:])
    else:
        all_(*[int(a) for a in sys.argv[2:5]])
