"""Bit-reproducibility of the hidden-256 step at configs[4]'s real size (1 object x 8192 rays x 128 samples), the check of
tests/test_fp16_gpu.py::test_hidden256_step_is_bit_reproducible_at_full_size with a report instead of an assertion: N
runs in one workspace + one in a fresh, POISONED workspace; per run, which gradient tensors differ from run 0, in how
many elements, by how much.  Any library build: OBJNERF_LIB=...; OBJ256_FIRST_FORM=1 selects the first form of kernel A.

    python3 tools/repro256.py [--modes bf16 fp16] [--feat 0 1] [--runs 4] [--rays 8192] [--poison 127]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
from openobj_amd import init as obj_init, ops, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", nargs="+", default=["bf16", "fp16"])
    ap.add_argument("--feat", nargs="+", type=int, default=[0, 1])
    ap.add_argument("--runs", type=int, default=4)
    ap.add_argument("--rays", type=int, default=8192)
    ap.add_argument("--objects", type=int, default=1)
    ap.add_argument("--poison", type=int, nargs="+", default=[0x7f, 0x00, 0xff])
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    K, R, n1, n2, H = args.objects, args.rays, 32, 96, 256
    bad = 0
    for mode in args.modes:
        for feat in args.feat:
            feat = bool(feat)
            arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
            arena.load_stacked(obj_init.init_stacked(K, H, 512, seed=47))
            b = synthetic.random_batch(K, R, n1, n2, seed=23, feat_dim=512 if feat else 0)
            batch = {k: torch.as_tensor(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])}
            ws = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, precision=mode)
            runs = []
            for i in range(args.runs):
                ws.grads.fill_(float(i))
                ops.train_step(arena, ws, batch, with_feat=feat, bf16=mode)
                torch.cuda.synchronize()
                runs.append(("same workspace %d" % i, ws.grads.clone(), ws.loss_terms.clone()))
            for pz in args.poison:
                ws2 = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, precision=mode)
                ws2.buf.fill_(pz)
                ws2.grads.fill_(float("nan"))
                ops.train_step(arena, ws2, batch, with_feat=feat, bf16=mode)
                torch.cuda.synchronize()
                runs.append(("fresh workspace filled with 0x%02x" % pz, ws2.grads.clone(), ws2.loss_terms.clone()))
                del ws2
            g0 = arena.views(runs[0][1])
            for name, g, t in runs[1:]:
                diffs = []
                for i, (v0, v) in enumerate(zip(g0, arena.views(g))):
                    if not feat and i in ops.FEAT_TENSORS:      # no gradient without gt_feat: the step leaves these entries alone
                        continue
                    ne = (v0 != v) | (v0.isnan() != v.isnan())
                    if bool(ne.any()):
                        diffs.append("%s %d/%d max %.3g" % (ops.TENSOR_NAMES[i], int(ne.sum()), ne.numel(),
                                                            float((v0 - v).abs().nan_to_num(float("inf")).max())))
                lt = "" if torch.equal(t, runs[0][2]) else " | loss terms differ %s vs %s" % (t.tolist(), runs[0][2].tolist())
                ok = not diffs and not lt
                bad += 0 if ok else 1
                print("%-4s feat=%d %-36s %s" % (mode, feat, name, "bit-equal" if ok else "DIFFERS: " + "; ".join(diffs) + lt), flush=True)
            del ws, arena
            torch.cuda.empty_cache()
    print("irreproducible runs:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
