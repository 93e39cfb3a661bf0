#!/usr/bin/env python3
"""Throughput bench of the hot path: training rays/s of the vectorised object-NeRF iteration.

    python bench.py --gpus N --steps K --warmup W [--config c2|c3|c4|c5]

One "step" = one complete training iteration of train.py:394-474: the stacked object networks of THIS rank (label
statistics -> fused forward / loss / backward kernel -> gradient finalize -> AdamW) plus the shared background network
(hidden 128, its 1200 rays split over the ranks), inputs already resident in HBM.  `value` counts object rays only.

Workloads (BASELINE.json configs; SURVEY.md 8 shape table):
  c2 (default)  50 objects per GPU, hidden 32, 4096 rays x 64 samples (16 + 48), RGB + depth + opacity loss   weak
  c3            c2 + the 512-d feature-distillation loss (cfg.part_mode)                                         weak
  c4            ScanNet-shaped: 120 objects IN TOTAL with the feature loss, sharded over the GPUs (15 each at 8)  strong
  c5            512-object stress: 64 objects per GPU, hidden 256, 8192 rays x 128 samples (32 + 96)              weak
Objects shard across ranks with no data-path collective.  An iteration has exactly two collectives
(openobj_amd.train.ShardedIteration): one int32[4] SUM before the step (early-return flags + background mask counts)
and one fp32 SUM of the replicated background network's gradient (182 339 floats + 4 loss terms) that is in flight
under the object kernel.

dtype: the headline line is fp32 -- the reference's arithmetic (train.py:74, AMP off) and the only mode held to the
1e-4 parity bar.  BASELINE.json's configs name bf16: the opt-in bf16-operand mode of the same step (fp32 accumulation,
master weights, compositing and AdamW; PSNR-gated) is timed in the same run and reported as the `bf16_mode` object
(or as the headline with --dtype bf16).

Prints ONE JSON line (rank 0): `roofline` = the dominant kernel against the dense fp32 MFMA peak (spec and measured),
`cpu_baseline` = the oracle (the reference's op sequence in PyTorch on the host cores) at BASELINE.md section 3's
shapes, `psnr` = the reconstruction quality of the same kernels on the G9 scene against the reference's ensemble.
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
PEAK_HBM_GBS = 8000.0             # HBM3E, MI355X_MICROARCH.md
VALU_SIMDS = 256 * 4              # SIMDs of the chip
VALU_CYCLES_PER_INST = 2.0        # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (cycle-constants table)
CLOCK_HZ = 2.4e9
RECORDED = os.path.join(ROOT, "profiles", "r02_counters.json")


def flop_per_ray(S: int, H: int = 32, feat: bool = False) -> float:
    """Algorithmic training FLOP per ray, SURVEY.md section 8(d): 3 * 2 * (S * M_s [+ 512 H])."""
    ms = 63 + (5 * H * H + 262 * H if feat else 4 * H * H + 220 * H)
    return 3.0 * 2.0 * (S * ms + (512 * H if feat else 0))


def recorded_counters(kernel: str, K, R, S):
    """PMC figures of `kernel` for this workload from the profile passes committed under profiles/ (rocprofv3 --pmc in
    separate runs, gfx950 FETCH_SIZE correction applied: tools/gpu_profile_round.sh).  Hardware counters cannot be
    read from inside this process: these are RECORDED values of the same launch, None when the workload differs."""
    try:
        with open(RECORDED) as f:
            for e in json.load(f)["kernels"]:
                if e["kernel"] == kernel and (e["objects"], e["rays"], e["samples"]) == (K, R, S):
                    return e
    except (OSError, KeyError, ValueError):
        pass
    return None


def measured_mfma_peak(dev, dtype_id):
    """TFLOP/s of a saturated MFMA loop (objnerf_mfma_peak: one wave per SIMD, four independent accumulators, operands
    in registers, non-trivial data) on THIS device -- the denominator a kernel can actually reach."""
    from openobj_amd import _lib
    n_wg, iters = 1024, 20000
    sink = torch.empty(n_wg * 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for it in (2000, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().objnerf_mfma_peak(dtype_id, it, n_wg, sink.data_ptr(), st), "objnerf_mfma_peak")
        e1.record()
        torch.cuda.synchronize()
    flop = (2048.0 if dtype_id == 0 else 16384.0) * 4 * iters * 4 * n_wg
    return flop / (e0.elapsed_time(e1) * 1e-3) / 1e12


def cpu_step_time(K, R, n1, n2, seed, steps, feat):
    """ms per step of the oracle = the reference's op sequence (vmap(pe) -> vmap(fc) -> step_batch_loss -> backward
    -> AdamW), fp32, on the host cores: 1 warm-up + `steps` timed, median."""
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
    return float(np.median(times[1:])) * 1e3


def cpu_baseline(feat, seed=4242):
    """c1 (K=1, R=256, S=32) in full; the 50-object stack at the reference-native R=120, S=10
    and at R=1024, S=64 (the metric's sample count); >= 3 timed steps after one warm-up.  `value` is the S=64 shape."""
    shapes = [("c1", 1, 256, 8, 24, 5), ("c2/c3 native", 50, 120, 1, 9, 3), ("c2/c3 metric", 50, 1024, 16, 48, 3)]
    rows = []
    for name, K, R, n1, n2, steps in shapes:
        f = feat and K > 1
        ms = cpu_step_time(K, R, n1, n2, seed, steps, f)
        rows.append({"shape": f"{name}: K={K} R={R} S={n1 + n2}{' +feat' if f else ''}", "ms_per_step": ms,
                     "rays_per_s": K * R / (ms * 1e-3), "timed_steps": steps})
    return dict(value=rows[-1]["rays_per_s"], unit="rays/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{rows[-1]['shape']}, hidden 32, {rows[-1]['timed_steps']} timed steps after 1 warm-up, "
                       f"{rows[-1]['ms_per_step']:.0f} ms/step (oracle/objnerf_oracle.py, torch {torch.__version__} CPU fp32)",
                shapes=rows)


def psnr_block(dev, n_seeds, with_bf16):
    """PSNR of the G9 scene after 300 fused iterations, ensemble over weight seeds, against the reference's own
    ensemble for the same seeds (tests/golden/g9_ensemble.npz)."""
    from openobj_amd import psnr_scene
    ref = psnr_scene.reference_ensemble()
    if ref is None:
        return None
    n = min(n_seeds, len(ref["seeds"]))
    seeds = [int(s) for s in ref["seeds"][:n]]
    sc = psnr_scene.PsnrScene(dev)
    out = {"scene": "G9: 4 analytic ellipsoids, 96 rays x 16 samples per object and iteration, 300 iterations, "
                    "PSNR of the rendered colour on 256 held-out rays per object",
           "reference": "the reference's own modules, same seeds (tests/golden/g9_ensemble.npz)"}
    out["f32"] = psnr_scene.delta_report(sc.ensemble(seeds, bf16=False), ref["psnr"][:n])
    if with_bf16:
        out["bf16"] = psnr_scene.delta_report(sc.ensemble(seeds, bf16=True), ref["psnr"][:n])
    return out


CONFIGS = {
    "c2": dict(objects=50, rays=4096, n1=16, n2=48, hidden=32, feat=False, scaling="weak"),
    "c3": dict(objects=50, rays=4096, n1=16, n2=48, hidden=32, feat=True, scaling="weak"),
    "c4": dict(objects=120, rays=4096, n1=16, n2=48, hidden=32, feat=True, scaling="strong"),
    "c5": dict(objects=64, rays=8192, n1=32, n2=96, hidden=256, feat=False, scaling="weak"),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json configuration (default c2; the explicit shape flags below override it)")
    ap.add_argument("--objects", type=int, default=None, help="object networks per GPU (c4: in total)")
    ap.add_argument("--rays", type=int, default=None, help="rays per object per step")
    ap.add_argument("--n-cam2surf", type=int, default=None)
    ap.add_argument("--n-bins", type=int, default=None)
    ap.add_argument("--hidden", type=int, default=None,
                    help="hidden width of the object networks (32 = the fused kernels; other widths run the layer-wise "
                         "path in object chunks)")
    ap.add_argument("--feat", action="store_true", default=None,
                    help="add the 512-d feature-distillation loss (cfg.part_mode)")
    ap.add_argument("--no-bg", dest="bg", action="store_false",
                    help="skip the shared background network.  Default (do_bg = 1, room_0.json:21): every step also "
                         "trains it (hidden 128, n_per_optim_bg = 1200 rays split over the ranks, same samples per ray, "
                         "gradient all-reduce over RCCL), as train.py:447-463 does; `value` counts object rays only")
    ap.add_argument("--dtype", choices=["f32", "bf16", "fp16"], default="f32",
                    help="f32 = the reference's arithmetic (default, the headline line).  bf16 = opt-in mode: MFMA "
                         "operands rounded to bf16, fp32 accumulation / master weights / compositing / AdamW")
    ap.add_argument("--no-bf16-line", dest="bf16_line", action="store_false",
                    help="do not also time the bf16 mode (reported as the `bf16_mode` object of the fp32 line)")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="diagnostic: background chain on the object kernel's stream instead of beside it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--psnr-seeds", type=int, default=128)
    ap.add_argument("--no-peak", action="store_true", help="skip the saturated-MFMA measurement")
    args = ap.parse_args()

    cfgname = args.config or "c2"
    wl = dict(CONFIGS[cfgname])
    for k_arg, k_wl in (("objects", "objects"), ("rays", "rays"), ("n_cam2surf", "n1"), ("n_bins", "n2"),
                        ("hidden", "hidden"), ("feat", "feat")):
        if getattr(args, k_arg) is not None:
            wl[k_wl] = getattr(args, k_arg)

    # stdout carries exactly ONE line (the JSON, rank 0).  Libraries print there too (RCCL's version banner at the
    # first collective): file descriptor 1 is pointed at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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

    from openobj_amd import cfg as ocfg, dist as odist, optim as ooptim, trainer as otrainer, train as otrain
    R, n1, n2, Hd, feat = wl["rays"], wl["n1"], wl["n2"], wl["hidden"], bool(wl["feat"])
    S = n1 + n2
    if wl["scaling"] == "strong":                       # a fixed population of objects dealt to the ranks
        lo, hi = odist.shard_objects(wl["objects"], world, rank)
        K, K_total = hi - lo, wl["objects"]
    else:
        K, K_total = wl["objects"], wl["objects"] * world
    bf16 = {"f32": False, "bf16": True, "fp16": "fp16"}[args.dtype]      # ops.precision_bits

    arena = ops.ParamArena(K, ops.NetShape(Hd, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, Hd, 512, seed=1000 + rank))
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    nb = 2 if Hd == 32 else 1       # resident batches, alternated so no step re-reads its own outputs
    batches = []
    for i in range(nb):
        src_k = K if Hd == 32 else min(K, 8)            # (the stress shape re-uses 8 objects' rays for its 64 networks)
        b = synthetic.random_batch(src_k, R, n1, n2, seed=4242 + 17 * rank + i, feat_dim=512 if feat else 0)
        reps = (K + src_k - 1) // src_k
        batches.append({k: torch.from_numpy(b[k]).to(dev).repeat(reps, *([1] * (b[k].ndim - 1)))[:K].contiguous()
                        for k in keys})

    class ObjLoop:                  # the object stack of this rank: fused step + AdamW over the arena
        def __init__(self):
            self.ws = ops.TrainWorkspace(arena, K, R, S, feat, precision=bf16)
            self.opt = ooptim.ArenaAdamW(arena, lr=1e-3, weight_decay=0.013)
            self.mask = arena.has_grad_mask(feat)
            self.bf16 = False

        def step(self, batch, global_flags=None):
            ops.train_step(arena, self.ws, batch, global_flags=global_flags, with_feat=feat, bf16=self.bf16)
            self.opt.step(self.ws.grads, self.mask, flags=global_flags if global_flags is not None else self.ws.flags)
            return self.ws.loss_terms

    obj_loop = ObjLoop()
    bg_loop, bg_batches = None, [None, None]
    if args.bg:
        c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev)))
        c.obj_id, c.hidden_feature_size, c.obj_scale = 0, c.hidden_feature_size_bg, c.bg_scale
        torch.manual_seed(7)                               # identical replica on every rank
        bg_loop = otrain.BackgroundLoop(c, otrainer.Trainer(c), with_feat=feat)
        lo, hi = odist.shard_rays(c.n_per_optim_bg, world, rank)
        bg_batches = []
        for i in range(2):
            b = synthetic.random_batch(1, c.n_per_optim_bg, n1, n2, seed=777 + i, feat_dim=512 if feat else 0)
            bg_batches.append({k: torch.from_numpy(b[k][:, lo:hi]).contiguous().to(dev) for k in keys})
    iteration = otrain.ShardedIteration(obj_loop, bg_loop, overlap=args.overlap, resident=True)   # (batches made above)

    def step(i, use_bf16):
        obj_loop.bf16 = use_bf16
        if bg_loop is not None:
            bg_loop.bf16 = use_bf16              # the mode applies to the whole step
        iteration.step(batches[i % nb], bg_batches[i & 1])

    def timed(use_bf16):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.  Then the dominant
        kernel alone: HIP events on the launch stream (torch's current stream, which the C ABI is handed) around
        objnerf_train_step = fused kernel + its slab reduction (the reduction is < 1 % of it)."""
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
        nk = max(3, min(args.steps, 20))
        ev0.record()
        for i in range(nk):
            ops.train_step(arena, obj_loop.ws, batches[i % nb], with_feat=feat, bf16=use_bf16)
        ev1.record()
        torch.cuda.synchronize()
        return dt_, ev0.elapsed_time(ev1) / nk

    dt, kern_ms = timed(bf16)
    status = int(obj_loop.ws.status.item())
    # the opt-in bf16-operand mode beside the fp32 headline (same step, same batches)
    bf16_extra = None
    if not bf16 and args.bf16_line:
        bf16_extra = timed(True)

    if rank == 0:
        rays_per_step = K_total * R if wl["scaling"] == "strong" else K * R * world
        value = rays_per_step * args.steps / dt
        fpr = flop_per_ray(S, H=Hd, feat=feat)
        fused = Hd == 32 and S <= 64 and bf16 != "fp16"
        k32 = "train_fused32_kernel<%s, false, %d>" % ("true" if feat else "false", 64 if S == 64 else 0)
        kbf = "train_fused_bf16_kernel<%s, %d>" % ("true" if feat else "false", 64 if S == 64 else 0)
        kname = (kbf if bf16 else k32) if fused else "objnerf_train_step, layer-wise path (batched MFMA GEMMs)"
        achieved = K * R * fpr / (kern_ms * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if bf16 else PEAK_F32_MFMA_TFLOPS
        rec = recorded_counters(kname, K, R, S)
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": rec["hbm_bytes_per_launch"] if rec else None,
                "traffic_note": ("RECORDED, not measured by this run: " + rec["source"]) if rec else
                                "no recorded PMC pass for this workload",
                "kernel": kname, "kernel_ms": kern_ms, "flop_per_ray": fpr,
                "algorithmic_bytes_per_launch": K * R * (S * 16 + 17 + (2048 * 2 if feat else 0))}
        if not fused and bf16 and Hd == 256 and not feat:
            # configs[4] in the 16-bit modes: every kernel of the layer-wise chain streams its operands once and is
            # HBM-bound.  Algorithmic bytes per sample: activations and back-propagated
            # gradients 2 H bytes each per pass, the fp32 embedding rows / their gradient as stored.
            bps = (1408 + 22 * Hd) + (2604 + 60 * Hd)
            gbs = K * R * S * bps / (kern_ms * 1e-3) / 1e9
            roof.update({"bound": "hbm", "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS,
                         "algorithmic_bytes_per_launch": K * R * S * bps, "bytes_per_sample": bps,
                         "mfma_tflops": achieved, "frac_of_bf16_mfma_peak": achieved / peak,
                         "note": "layer-wise chain, 16-bit activations and gradients; bytes per sample = forward "
                                 "1408 + 22 H, backward 2604 + 60 H (each buffer counted once per kernel that streams it)"})
        if not args.no_peak:
            pm = measured_mfma_peak(dev, 1 if bf16 else 0)
            roof["peak_measured"] = pm
            roof["frac_of_measured_peak"] = achieved / pm          # (of the measured MFMA peak)
        out = {
            "metric": "training rays/sec/GPU @64 samples/ray, 50 obj; PSNR delta vs ref",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": wl["scaling"],
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{cfgname}: Replica room_0-shaped, {K} object MLPs on this GPU"
                                   f"{' of %d in total' % K_total if wl['scaling'] == 'strong' else ''} (hidden {Hd}), "
                                   f"{R} rays/object/step, {S} samples/ray ({n1}+{n2}), RGB+depth+opacity"
                                   f"{'+512-d feature' if feat else ''} loss, fused fwd+loss+bwd+AdamW",
                       "objects_per_gpu": K, "objects_total": K_total, "rays_per_object": R, "samples_per_ray": S,
                       "hidden": Hd, "feature_head": feat, "background_mlp": bool(args.bg),
                       "parallelism": f"objects sharded x{world}", "collectives_per_step": 2 if use_dist else 0,
                       "loss_status": status},
            "rays_per_sec_per_gpu": value / world,
            "roofline": roof,
        }
        if bf16_extra is not None:
            bdt, bk = bf16_extra
            bkname = kbf if fused else "layer-wise path, bf16-operand GEMMs"
            mf = K * R * fpr / (bk * 1e-3) / 1e12
            brec = recorded_counters(bkname, K, R, S)
            broof = {"bound": "valu", "unit": "wave instructions", "kernel": bkname, "kernel_ms": bk,
                     "mfma_tflops": mf, "frac_of_bf16_mfma_peak": mf / PEAK_BF16_MFMA_TFLOPS}
            if brec:
                floor_ms = brec["valu_insts_per_launch"] * VALU_CYCLES_PER_INST / (VALU_SIMDS * CLOCK_HZ) * 1e3
                broof.update({"valu_insts_per_launch": brec["valu_insts_per_launch"],
                              "valu_insts_per_sample": brec["valu_insts_per_launch"] * 64.0 / (K * R * S),
                              "valu_floor_ms": floor_ms, "frac": floor_ms / bk,
                              "note": "VALU issue floor = wave instructions x 2 cycles / (1024 SIMDs x 2.4 GHz); "
                                      "instruction count RECORDED: " + brec["source"]})
            out["bf16_mode"] = {"value": rays_per_step * args.steps / bdt, "unit": "rays/s",
                                "ms_per_step": bdt / args.steps * 1e3, "roofline": broof,
                                "note": "OBJNERF_TRAIN_BF16: bf16 MFMA operands, fp32 accumulate / master weights / "
                                        "compositing / AdamW; PSNR-gated (tests/test_bf16_gpu.py), not 1e-4 parity"}
        if world == 1 and not args.no_psnr and Hd == 32:
            ps = psnr_block(dev, args.psnr_seeds, with_bf16=args.bf16_line or bf16)
            if ps is not None:
                out["psnr"] = ps
                key = "bf16" if bf16 else "f32"
                out["psnr_delta_db"] = ps[key]["delta_db"]
                out["psnr_delta_ci95_db"] = ps[key]["ci95_db"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(feat)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
