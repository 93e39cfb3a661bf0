"""What the paired hidden-256 PSNR gate detects (tests/test_psnr_gpu.py::test_hidden_256_network_psnr_paired_early).

The same 33-seed runs as the gate (fixture G9D: the reference's own modules after 10 / 20 iterations), with the gradients
DEGRADED between the step and AdamW: g <- g + h * rms(g over the tensor, per object) * N(0, 1), seeded.  (Adam normalises
a gradient's scale away, so a handicap has to be noise relative to the tensor's own magnitude; h = 0 is the gate itself.)
For each mode and strength: the paired report and psnr_scene.paired_gate_failures -- the record shows the strength at
which the mean PSNR after 10 iterations has dropped by ~0.3 dB and that the gate is red there.

    python3 tools/h256_handicap.py [--modes fp16 bf16 f32] [--strengths 0 0.25 0.5 1 2]  > profiles/r06_h256_handicap.txt
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import psnr_scene  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", nargs="+", default=["fp16", "bf16", "f32"])
    ap.add_argument("--strengths", nargs="+", type=float, default=[0.0, 0.25, 0.5, 1.0, 2.0])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    ref = psnr_scene.reference_early_d()
    seeds = [int(x) for x in ref["seeds"]]
    er = psnr_scene.EnsembleRun(dev, with_feat=False, spec=dict(psnr_scene.G9C, steps=20, early=10))
    for mode in args.modes:
        for h in args.strengths:
            gen = torch.Generator(device=dev).manual_seed(1234)

            def hook(it, grads, h=h, gen=gen):
                if h == 0.0:
                    return
                views = er_arena_views(grads)
                for v in views:                                       # [nK, ...] per tensor
                    rms = v.reshape(v.shape[0], -1).pow(2).mean(1).sqrt().reshape([-1] + [1] * (v.dim() - 1))
                    v.add_(h * rms * torch.randn(v.shape, device=dev, generator=gen))

            run = er.run(seeds, psnr_scene.MODES[mode], grad_hook=hook)
            r10 = psnr_scene.paired_report(run["psnr50"], ref["psnr10"])
            r20 = psnr_scene.paired_report(run["psnr300"], ref["psnr20"])
            bad = psnr_scene.paired_gate_failures(mode, r10, r20)
            print("%-4s handicap %.2f: 10 it mean %+.3f +- %.3f max %.2f | 20 it mean %+.3f max %.2f | gate %s %s"
                  % (mode, h, r10["mean_delta_db"], r10["ci95_db"], r10["max_abs_delta_db"], r20["mean_delta_db"],
                     r20["max_abs_delta_db"], "RED" if bad else "green", "; ".join(bad)), flush=True)


def er_arena_views(grads):
    """The per-tensor views [nK, ...] of a flat gradient buffer laid out like the arena (hidden 256, 512-d head)."""
    from openobj_amd import ops
    global _ARENA
    if "_ARENA" not in globals() or _ARENA.K != grads.shape[0]:
        _ARENA = ops.ParamArena(grads.shape[0], ops.NetShape(256, 512, 6), grads.device)
    return [v for i, v in enumerate(_ARENA.views(grads)) if i not in ops.FEAT_TENSORS]


if __name__ == "__main__":
    main()
