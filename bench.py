#!/usr/bin/env python3
"""Throughput bench of the hot path: training rays/s of the vectorised object-NeRF iteration.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete training iteration of train.py:394-474: the K_obj stacked object networks of
THIS rank (label statistics -> fused forward/loss/backward kernel -> gradient finalize -> AdamW) plus
the shared background network (hidden 128, its 1200 rays split over the ranks, gradient all-reduce),
inputs already resident in HBM.  Workload (BASELINE.json configs[1]): 50 objects per GPU,
hidden 32, 64 samples per ray (n_bins_cam2surface 16 + n_bins 48), RGB + depth + opacity loss,
synthetic Replica-shaped rays (openobj_amd.synthetic.random_batch), reference-initialised weights.
Objects shard across ranks with no data-path collective (the per-step early-return flags are a
2-int all-reduce); the replicated background network's gradient (182 339 floats) is the one RCCL
all-reduce.  Scaling is weak: every rank trains its own 50 objects.  `value` counts object rays only.

dtype: the headline line is fp32 -- the reference's arithmetic (train.py:74, AMP off) and the only mode held to
the 1e-4 parity bar.  BASELINE.json configs[1] names bf16: the opt-in bf16-operand mode of the same kernel
(fp32 accumulation, master weights, compositing and AdamW; PSNR-gated) is timed in the same run and reported
as the `bf16_mode` object of the line (or as the headline with --dtype bf16).

Prints ONE JSON line (rank 0).  `roofline` is the fused kernel against the dense fp32 MFMA peak,
`cpu_baseline` is the oracle (the reference's op sequence in PyTorch on the host cores) on a bounded
sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from openobj_amd import init as obj_init  # noqa: E402
from openobj_amd import ops, synthetic    # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 MFMA


def flop_per_ray(S: int, H: int = 32, feat: bool = False) -> float:
    """Algorithmic training FLOP per ray, SURVEY.md section 8(d): 3 * 2 * (S * M_s [+ 512 H])."""
    ms = 63 + (5 * H * H + 262 * H if feat else 4 * H * H + 220 * H)
    return 3.0 * 2.0 * (S * ms + (512 * H if feat else 0))


def measured_traffic(K, R, S, feat, dtype):
    """HBM bytes per launch of the fused kernel from the PMC passes committed under profiles/ (FETCH_SIZE and
    WRITE_SIZE in separate rocprofv3 --pmc runs, gfx950 correction applied; tools/gpu_profile_round.sh).
    Counters cannot be read from inside this process, so the figure is the recorded one for the same workload;
    None when the workload differs from the profiled one."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic_v5.json")) as f:
            t = json.load(f)
        wl = t["workload"]
        if (wl["objects"], wl["rays"], wl["samples"], wl["feat"], wl["dtype"]) == (K, R, S, feat, dtype):
            return t["traffic_bytes_per_launch"]
    except (OSError, KeyError, ValueError):
        pass
    return None


def cpu_baseline(K, R, n1, n2, seed, steps=2, feat=False):
    """Oracle = the reference's op sequence (vmap(pe) -> vmap(fc) -> step_batch_loss -> backward -> AdamW),
    fp32, on the host cores.  Bounded sample: same K, S, H; fewer rays per object."""
    from oracle import objnerf_oracle as O
    stacked = obj_init.init_stacked(K, 32, 512, seed=0)
    fc = [p.clone().requires_grad_(True) for p in stacked[:18]]
    B = stacked[18].clone().requires_grad_(True)
    params = fc + [B]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    b = synthetic.random_batch(K, R, n1, n2, seed=seed, feat_dim=512 if feat else 0)
    tb = {k: torch.from_numpy(b[k]) for k in ["pts", "gt_depth", "gt_rgb", "labels", "z"] + (["gt_feat"] if feat else [])}
    scale = torch.full((K,), 2.0)
    times = []
    for it in range(steps + 1):
        t0 = time.perf_counter()
        loss, _ = O.train_forward_loss(fc, B, scale, tb["pts"], tb["gt_depth"], tb["gt_rgb"], tb["labels"], tb["z"],
                                       gt_feat=tb["gt_feat"] if feat else None)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                if g is not None:
                    O.adamw_step(p, g, mm, vv, it + 1, 1e-3, 0.013)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return dict(value=K * R / t, unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"K={K} objects x R={R} rays x S={n1 + n2} samples, hidden 32, {steps} timed steps, "
                       f"{t * 1e3:.0f} ms/step (oracle/objnerf_oracle.py, torch {torch.__version__} CPU fp32)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--objects", type=int, default=50, help="object networks per GPU")
    ap.add_argument("--rays", type=int, default=4096, help="rays per object per step")
    ap.add_argument("--n-cam2surf", type=int, default=16)
    ap.add_argument("--n-bins", type=int, default=48)
    ap.add_argument("--hidden", type=int, default=32,
                    help="hidden width of the object networks (32 = the fused kernel; other widths, e.g. BASELINE "
                         "configs[4] hidden 256 with --n-cam2surf 32 --n-bins 96, run the layer-wise path in object chunks)")
    ap.add_argument("--feat", action="store_true",
                    help="BASELINE configs[2]: add the 512-d feature-distillation loss (cfg.part_mode)")
    ap.add_argument("--no-bg", dest="bg", action="store_false",
                    help="skip the shared background network.  Default (do_bg = 1, room_0.json:21): every step also "
                         "trains it (hidden 128, n_per_optim_bg = 1200 rays split over the ranks, 64 samples/ray, "
                         "gradient all-reduce over RCCL), as train.py:447-463 does; `value` counts object rays only")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 = the reference's arithmetic (default, the headline line).  bf16 = opt-in mode: MFMA "
                         "operands rounded to bf16, fp32 accumulation / master weights / compositing / AdamW")
    ap.add_argument("--no-bf16-line", dest="bf16_line", action="store_false",
                    help="do not also time the bf16 mode (reported as the `bf16_mode` object of the fp32 line)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rays", type=int, default=192)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    use_dist = world > 1 or (os.environ.get("OBJNERF_DIST_SELFTEST") == "1" and "RANK" in os.environ)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    K, R, n1, n2 = args.objects, args.rays, args.n_cam2surf, args.n_bins
    S = n1 + n2
    Hd = args.hidden
    arena = ops.ParamArena(K, ops.NetShape(Hd, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, Hd, 512, seed=1000 + rank))
    feat = bool(args.feat)
    bf16 = args.dtype == "bf16"
    ws = ops.TrainWorkspace(arena, K, R, S, feat)
    m = torch.zeros_like(arena.params)
    v = torch.zeros_like(arena.params)
    mask = arena.has_grad_mask(feat)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    batches = []
    for i in range(2):      # two resident batches, alternated, so no step re-reads its own outputs
        b = synthetic.random_batch(K, R, n1, n2, seed=4242 + 17 * rank + i, feat_dim=512 if feat else 0)
        batches.append({k: torch.from_numpy(b[k]).to(dev) for k in keys})
    gflags = torch.zeros(2, dtype=torch.int32, device=dev)
    bg_loop = None
    if args.bg:
        from openobj_amd import cfg as ocfg, dist as odist, trainer as otrainer, train as otrain
        c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev)))
        c.obj_id, c.hidden_feature_size, c.obj_scale = 0, c.hidden_feature_size_bg, c.bg_scale
        torch.manual_seed(7)                               # identical replica on every rank
        bg_loop = otrain.BackgroundLoop(c, otrainer.Trainer(c), with_feat=feat)
        lo, hi = odist.shard_rays(c.n_per_optim_bg, world, rank)
        bg_batches = []
        for i in range(2):
            b = synthetic.random_batch(1, c.n_per_optim_bg, n1, n2, seed=777 + i, feat_dim=512 if feat else 0)
            bg_batches.append({k: torch.from_numpy(b[k][:, lo:hi]).contiguous().to(dev) for k in keys})

    step_no = [0]

    def step(i, use_bf16):
        from openobj_amd import _lib
        b = batches[i & 1]
        if use_dist:
            # the early return of render_rays.py:89-94 spans every object of the batch -> global flags
            _lib.check(_lib.lib().objnerf_label_counts(K, R, b["labels"].data_ptr(), ws.counts.data_ptr(),
                                                      gflags.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "label_counts")
            dist.all_reduce(gflags, op=dist.ReduceOp.MAX)
            ops.train_step(arena, ws, b, global_flags=gflags, with_feat=feat, bf16=use_bf16)
        else:
            ops.train_step(arena, ws, b, with_feat=feat, bf16=use_bf16)
        step_no[0] += 1
        ops.adamw_step(arena, ws.grads, m, v, mask, step_no[0], 1e-3, 0.013)
        if bg_loop is not None:
            bg_loop.bf16 = use_bf16              # the mode applies to the whole step
            bg_loop.step(bg_batches[i & 1])

    def timed(use_bf16):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.  Then the
        dominant kernel alone: HIP events on the launch stream around objnerf_train_step (fused kernel +
        finalize; the finalize is <1 % of it)."""
        for i in range(args.warmup):
            step(i, use_bf16)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i, use_bf16)
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([dt_], device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        ev0 = torch.cuda.Event(enable_timing=True)
        ev1 = torch.cuda.Event(enable_timing=True)
        nk = max(5, min(args.steps, 20))
        ev0.record()
        for i in range(nk):
            ops.train_step(arena, ws, batches[i & 1], with_feat=feat, bf16=use_bf16)
        ev1.record()
        torch.cuda.synchronize()
        return dt_, ev0.elapsed_time(ev1) / nk

    dt, kern_ms = timed(bf16)
    status = int(ws.status.item())
    # the opt-in bf16-operand mode beside the fp32 headline (same step, same batches)
    bf16_extra = None
    if not bf16 and args.bf16_line:
        bdt, bk = timed(True)
        bf16_extra = (bdt, bk)

    if rank == 0:
        rays_per_step = K * R * world
        value = rays_per_step * args.steps / dt
        fpr = flop_per_ray(S, H=Hd, feat=feat)
        achieved = K * R * fpr / (kern_ms * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        kname = (f"train_fused_bf16_kernel<{'true' if feat else 'false'}>" if bf16 else ("train_fused_kernel<true, false>" if feat else "train_fused32_kernel<false>"))
        if Hd != 32 or S > 64:
            kname = "objnerf_train_step, layer-wise path (batched MFMA GEMMs)"
        out = {
            "metric": "training rays/sec/GPU @64 samples/ray, 50 obj; PSNR delta vs ref",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"Replica room_0-shaped, {K} object MLPs/GPU (hidden {Hd}), {R} rays/object/step, "
                                   f"{S} samples/ray ({n1}+{n2}), RGB+depth+opacity"
                                   f"{'+512-d feature' if feat else ''} loss, fused fwd+loss+bwd+AdamW",
                       "objects_per_gpu": K, "rays_per_object": R, "samples_per_ray": S, "hidden": Hd,
                       "feature_head": feat, "background_mlp": bool(args.bg), "parallelism": f"objects sharded x{world}",
                       "loss_status": status},
            "rays_per_sec_per_gpu": value / world,
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "traffic": measured_traffic(K, R, S, feat, args.dtype),
                         "traffic_unit": "bytes/launch (PMC, profiles/r01_traffic_v5.json)",
                         "kernel": kname, "kernel_ms": kern_ms,
                         "flop_per_ray": fpr},
        }
        if bf16_extra is not None:
            bdt, bk = bf16_extra
            out["bf16_mode"] = {"value": rays_per_step * args.steps / bdt, "unit": "rays/s",
                                "ms_per_step": bdt / args.steps * 1e3,
                                "kernel": (f"train_fused_bf16_kernel<{'true' if feat else 'false'}>" if Hd == 32 and S <= 64
                                           else "layer-wise path, bf16-operand GEMMs"),
                                "kernel_ms": bk, "mfma_tflops": K * R * fpr / (bk * 1e-3) / 1e12,
                                "note": "OBJNERF_TRAIN_BF16: bf16 MFMA operands, fp32 accumulate / master weights / "
                                        "compositing / AdamW; PSNR-gated (tests/test_bf16_gpu.py), not 1e-4 parity"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(K, args.cpu_rays, n1, n2, seed=4242, feat=feat)
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
