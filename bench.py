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

Ranks.  `--gpus N` with N > 1 and no WORLD_SIZE in the environment makes THIS process a launcher: before anything
touches the GPU it starts N fresh `bench.py` processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in
their environment, one per GPU, RCCL between them), relays rank 0's single JSON line and exits non-zero if any rank
did.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the environment already names the
ranks and the process is one of them.  `--dry-launch` is the CPU form of the same launch (gloo, no kernels): the test
of the launcher and of the collectives' host path (tests/test_bench_launcher.py).

dtype: the headline line is fp32 -- the reference's arithmetic (train.py:74, AMP off) and the only mode held to the
1e-4 parity bar.  BASELINE.json's configs name bf16: the opt-in bf16-operand mode of the same step (fp32 accumulation,
master weights, compositing and AdamW; PSNR-gated) is timed in the same run and reported as the `bf16_mode` object
(or as the headline with --dtype bf16).

Prints ONE short JSON line (rank 0; < 4 KB, asserted: the driver parses it): the contract's keys, `roofline` = the dominant
launch against the dense MFMA peak of its operand type (spec and measured; `traffic` = the RECORDED PMC bytes of the same
launch, profiles/r06_counters.json), `cpu_baseline` = the oracle (the reference's op sequence in PyTorch on the host cores) at
BASELINE.md section 3's metric shape, the PSNR delta of the trained model, a flat `summary` (the other BASELINE configurations'
rays/s and ms per step, the native-shape frame) and `detail` = the path of the FULL report (--detail-out, default
gpurun_out/bench_detail.json): `other_configs` (c3, configs[3]'s per-GPU share, configs[4]'s per-GPU share in fp16 without /
with the feature loss, the mapping loop at the native shape), the `psnr` block, the per-shape CPU rows, `bf16_mode`.  A
distributed run adds `dist_selftest` = the iteration's two collectives timed on their own.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0    # dense bf16 / fp16 MFMA
PEAK_HBM_GBS = 8000.0             # HBM3E, MI355X_MICROARCH.md
VALU_SIMDS = 256 * 4              # SIMDs of the chip
VALU_CYCLES_PER_INST = 2.0        # a wave64 VALU instruction occupies its SIMD-32 for 2 cycles (cycle-constants table)
CLOCK_HZ = 2.4e9
RECORDED = os.path.join(ROOT, "profiles", "r06_counters.json")
HBM_SUSTAINED_GBS = 6300.0        # what streaming kernels sustain of the 8 TB/s (MI355X_MICROARCH.md: 6.0-6.3 TB/s)
BG_GRAD_FLOATS = 182339 + 4       # the replicated background network's gradient + its four loss terms (collective 2)

CONFIGS = {
    "c2": dict(objects=50, rays=4096, n1=16, n2=48, hidden=32, feat=False, scaling="weak"),
    "c3": dict(objects=50, rays=4096, n1=16, n2=48, hidden=32, feat=True, scaling="weak"),
    "c4": dict(objects=120, rays=4096, n1=16, n2=48, hidden=32, feat=True, scaling="strong"),
    "c5": dict(objects=64, rays=8192, n1=32, n2=96, hidden=256, feat=False, scaling="weak"),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)       # >= 20 warm-up + >= 100 timed steps
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default=None,
                    help="BASELINE.json configuration (default c2; the explicit shape flags below override it)")
    ap.add_argument("--objects", type=int, default=None, help="object networks per GPU (c4: in total)")
    ap.add_argument("--bg-ranks", type=int, default=None,
                    help="per-GPU share runs: this process stands for one of N ranks and takes one rank's slice of the "
                         "background rays (the default line's c4_share entry uses 8)")
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
    ap.add_argument("--no-pipeline", dest="pipelined", action="store_false",
                    help="make the object stream wait for the background chain at the end of EVERY step (round 4's form)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-psnr", action="store_true")
    ap.add_argument("--psnr-seeds", type=int, default=320)
    ap.add_argument("--no-peak", action="store_true", help="skip the saturated-MFMA measurement")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="default line only: do not also time c3, configs[3]'s share and configs[4]'s share")
    ap.add_argument("--other-steps", type=int, default=10, help="timed steps of each `other_configs` entry")
    ap.add_argument("--detail-out", default=None,
                    help="file the FULL report goes to (other_configs, psnr detail, native_frame, per-shape cpu rows, notes); "
                         "default gpurun_out/bench_detail.json.  stdout carries ONE short line (< 4 KB) naming this file")
    ap.add_argument("--share-curve-out", default=None,
                    help="also time the emulated 1 / 2 / 4 / 8-rank shares of c4 and c2 (fp32 and bf16, --steps each) and "
                         "write the table to this file (profiles/r05_share_curve.json)")
    ap.add_argument("--dry-launch", action="store_true",
                    help="CPU form of the N-rank launch: gloo instead of RCCL, the iteration's two collectives on host "
                         "buffers, no kernels; the line carries \"dry_launch\": true and value 0")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# launcher: `--gpus N` without an outer torchrun.  Runs before torch / the package are imported: nothing here may
# initialise the GPU (a process that has must not be the parent of the ranks' rendezvous, and is never re-exec'd).
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _device_count_in_child() -> int:
    """torch.cuda.device_count() read in a CHILD process (the launcher itself never asks the GPU runtime anything)."""
    r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    try:
        return int(r.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        sys.stderr.write("bench.py: cannot count GPUs: " + r.stderr[-2000:] + "\n")
        return 0


def launch_ranks(args, argv) -> int:
    n = args.gpus
    if not args.dry_launch:
        have = _device_count_in_child()
        if n > have:
            sys.stderr.write(f"bench.py: --gpus {n} but this node shows {have} GPU(s); refusing to oversubscribe\n")
            return 2
    env0 = dict(os.environ)
    env0.update({"WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                 "MASTER_PORT": str(_free_port()), "OBJNERF_BENCH_LAUNCHED": "1"})
    env0.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this pool
    cmd = [sys.executable, os.path.abspath(__file__)] + list(argv)
    procs = []
    for r in range(n):
        env = dict(env0)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r)})
        # rank 0's stdout carries the line; the other ranks print nothing there (their stdout goes to our stderr)
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno()))
    # supervise EVERY rank: the first one that exits non-zero ends the launch (a rank that dies at import or device
    # set-up would otherwise leave the others in the rendezvous until torch's own timeout); the whole launch is bounded.
    # Only the children started above are ever signalled; nothing is restarted.
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("OBJNERF_BENCH_LAUNCH_TIMEOUT", "3600"))
    codes = [None] * n
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        failed = any(c not in (None, 0) for c in codes)
        if failed or time.time() > deadline:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    p.terminate()
            t_kill = time.time() + 10
            for r, p in enumerate(procs):
                if codes[r] is None:
                    try:
                        codes[r] = p.wait(timeout=max(0.1, t_kill - time.time()))
                    except subprocess.TimeoutExpired:
                        p.kill()                              # (the exact child we started)
                        codes[r] = p.wait()
            if not failed:
                sys.stderr.write("bench.py: launch timed out\n")
                codes = [c if c != 0 else -1 for c in codes]
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    lines = [ln for ln in out0.decode(errors="replace").splitlines() if ln.strip().startswith("{")]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or len(lines) != 1:
        sys.stderr.write(f"bench.py: launch of {n} ranks failed: exit codes {codes}, {len(lines)} JSON line(s) from rank 0\n")
        return 1
    sys.stdout.write(lines[0] + "\n")
    sys.stdout.flush()
    return 0


# ----------------------------------------------------------------------------------------------------------------------
def flop_per_ray(S: int, H: int = 32, feat: bool = False) -> float:
    """Algorithmic training FLOP per ray, SURVEY.md section 8(d): 3 * 2 * (S * M_s [+ 512 H])."""
    ms = 63 + (5 * H * H + 262 * H if feat else 4 * H * H + 220 * H)
    return 3.0 * 2.0 * (S * ms + (512 * H if feat else 0))


def algorithmic_bytes(K, R, S, feat, scope="step"):
    """SURVEY.md 8(d), points / z supplied: 16 B per sample + 17 B per ray.  With the feature loss the STEP also reads the
    fp32 target feature once (2 KB per ray) -- but not the fused kernel: feat_pre_kernel / feat_post_kernel do, the fused
    kernel reads their 34-float record per ray and writes its own 36-float one.  scope = "kernel" is
    what the fused kernel's recorded FETCH / WRITE counters have to be compared with (round 4 compared them with the
    step figure: traffic ratios below 1)."""
    per_ray = S * 16 + 17
    if feat:
        per_ray += 2048 if scope == "step" else (34 + 36) * 4
    return K * R * per_ray


def recorded_counters(kernel: str, K, R, S):
    """PMC figures of `kernel` for this workload from the profile passes committed under profiles/ (rocprofv3 --pmc in
    separate runs, gfx950 FETCH_SIZE correction applied: tools/gpu_profile_round3.sh).  Hardware counters cannot be
    read from inside this process: these are RECORDED values of the same launch, None when the workload differs.
    Entries recorded per object (the hidden-256 path runs its objects in workspace chunks) are scaled to K objects."""
    for path in (RECORDED, RECORDED.replace("r06_", "r05_"), RECORDED.replace("r06_", "r04_"), RECORDED.replace("r06_", "r03_")):
        try:
            with open(path) as f:
                for e in json.load(f)["kernels"]:
                    if e["kernel"] != kernel or (e["rays"], e["samples"]) != (R, S):
                        continue
                    if e.get("per_object"):
                        e = dict(e)
                        for k in ("hbm_bytes_per_launch", "valu_insts_per_launch", "mfma_insts_per_launch"):
                            if e.get(k) is not None:
                                e[k] = e[k] * K
                        return e
                    if e["objects"] == K:
                        return e
        except (OSError, KeyError, ValueError):
            pass
    return None


def measured_mfma_peak(dev, dtype_id):
    """TFLOP/s of a saturated MFMA loop (objnerf_mfma_peak: one wave per SIMD, two independent 32x32 accumulator chains,
    operands in registers, non-trivial data) on THIS device -- the denominator a kernel can actually reach."""
    import torch
    from openobj_amd import _lib
    n_wg, iters = 1024, 10000
    sink = torch.empty(n_wg * 256, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for it in (2000, iters):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.check(_lib.lib().objnerf_mfma_peak(dtype_id, it, n_wg, sink.data_ptr(), st), "objnerf_mfma_peak")
        e1.record()
        torch.cuda.synchronize()
    flop = (4096.0 if dtype_id == 0 else 32768.0) * 4 * iters * 4 * n_wg
    return flop / (e0.elapsed_time(e1) * 1e-3) / 1e12


def cpu_step_time(K, R, n1, n2, seed, steps, feat):
    """ms per step of the oracle = the reference's op sequence (vmap(pe) -> vmap(fc) -> step_batch_loss -> backward
    -> AdamW), fp32, on the host cores: 1 warm-up + `steps` timed, median."""
    import numpy as np
    import torch
    from openobj_amd import init as obj_init, synthetic
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
    import torch
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


def psnr_block(dev, n_seeds, with_bf16, with_fp16=False):
    """PSNR half of the metric on the same kernels (openobj_amd.psnr_scene): the well-posed per-seed comparison after
    50 iterations and the ensemble comparison after 300, both against the reference's own modules on the same seeds
    (tests/golden/g9_ensemble*.npz)."""
    from openobj_amd import psnr_scene
    return psnr_scene.report(dev, n_seeds, modes=["f32"] + (["bf16"] if with_bf16 else []) + (["fp16"] if with_fp16 else []))


# ----------------------------------------------------------------------------------------------------------------------
class Workload:
    """One BASELINE configuration resident on this rank's GPU: arena, alternating batches, the object loop, the
    background loop and the sharded iteration around them."""

    def __init__(self, cfgname, wl, args, dev, world, rank, precision):
        import torch
        from openobj_amd import cfg as ocfg, dist as odist, init as obj_init, ops, optim as ooptim, synthetic
        from openobj_amd import trainer as otrainer, train as otrain
        self.name, self.wl, self.args, self.dev, self.world, self.rank = cfgname, wl, args, dev, world, rank
        R, n1, n2, Hd, feat = wl["rays"], wl["n1"], wl["n2"], wl["hidden"], bool(wl["feat"])
        self.R, self.n1, self.n2, self.Hd, self.feat, self.S = R, n1, n2, Hd, feat, n1 + n2
        if wl["scaling"] == "strong":                       # a fixed population of objects dealt to the ranks
            lo, hi = odist.shard_objects(wl["objects"], world, rank)
            K, K_total = hi - lo, wl["objects"]
        else:
            K, K_total = wl["objects"], wl["objects"] * world
        self.K, self.K_total = K, K_total
        self.precision = precision                            # ops.precision_bits: False / True ("bf16") / "fp16"
        self.ops = ops
        arena = self.arena = ops.ParamArena(K, ops.NetShape(Hd, 512, 6), dev)
        arena.load_stacked(obj_init.init_stacked(K, Hd, 512, seed=1000 + rank))
        keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
        self.nb = 2 if Hd == 32 else 1  # resident batches, alternated so no step re-reads its own outputs
        self.distinct = K                                    # every network trains on its own rays (round 5: configs[4] too)
        self.batches = []
        for i in range(self.nb):
            b = synthetic.random_batch(self.distinct, R, n1, n2, seed=4242 + 17 * rank + i, feat_dim=512 if feat else 0)
            reps = (K + self.distinct - 1) // self.distinct
            self.batches.append({k: torch.from_numpy(b[k]).to(dev).repeat(reps, *([1] * (b[k].ndim - 1)))[:K].contiguous()
                                 for k in keys})
        S = self.S

        class ObjLoop:                  # the object stack of this rank: fused step + AdamW over the arena
            def __init__(self):
                self.ws = ops.TrainWorkspace(arena, K, R, S, feat, precision=precision)
                self.opt = ooptim.ArenaAdamW(arena, lr=1e-3, weight_decay=0.013)
                self.mask = arena.has_grad_mask(feat)
                self.bf16 = False
                self.events = None              # timed(): one HIP event pair per step around objnerf_train_step

            def step(self, batch, global_flags=None):
                if self.events is not None:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                # (optim=: AdamW runs in the step's last launch -- objnerf_train_args.optim, ABI 7)
                ops.train_step(arena, self.ws, batch, global_flags=global_flags, with_feat=feat, bf16=self.bf16,
                               optim=self.opt)
                if self.events is not None:
                    e1.record()
                    self.events.append((e0, e1))
                return self.ws.loss_terms

        self.obj_loop = ObjLoop()
        self.bg_loop, self.bg_batches = None, [None, None]
        if args.bg and wl.get("bg", True):
            c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev)))
            c.obj_id, c.hidden_feature_size, c.obj_scale = 0, c.hidden_feature_size_bg, c.bg_scale
            torch.manual_seed(7)                               # identical replica on every rank
            self.bg_loop = otrain.BackgroundLoop(c, otrainer.Trainer(c), with_feat=feat)
            # `bg_ranks` (the c4_share entry): this process stands for ONE of that many ranks, so it takes one rank's
            # slice of the background rays as dist.shard_rays deals them (the gradient all-reduce is not simulated)
            self.bg_ranks = world * int(wl.get("bg_ranks", 1))
            lo, hi = odist.shard_rays(c.n_per_optim_bg, self.bg_ranks, rank)
            self.bg_batches = []
            for i in range(2):
                b = synthetic.random_batch(1, c.n_per_optim_bg, n1, n2, seed=777 + i, feat_dim=512 if feat else 0)
                self.bg_batches.append({k: torch.from_numpy(b[k][:, lo:hi]).contiguous().to(dev) for k in keys})
        # (pipelined: the background chain of step i may still run under the object kernel of step i + 1 -- both chains stay
        # in order on their own streams; timed() synchronises the device before it reads the clock)
        self.iteration = otrain.ShardedIteration(self.obj_loop, self.bg_loop, overlap=args.overlap, resident=True,
                                                 device=dev, pipelined=args.pipelined)

    def step(self, i, mode):
        self.obj_loop.bf16 = mode
        if self.bg_loop is not None:
            self.bg_loop.bf16 = mode             # the mode applies to the whole step
        self.iteration.step(self.batches[i % self.nb], self.bg_batches[i & 1])

    def timed(self, mode, steps, warmup, dist=None):
        """W warm-up steps, then exactly K steps between barrier + synchronize; max over ranks.  The dominant launch
        (objnerf_train_step = fused kernel + its slab reduction, < 1 % of it) is timed INSIDE the same K steps: one
        HIP event pair per step on the launch stream (torch's current stream, which the C ABI is handed), so
        `kernel_ms` <= `ms_per_step` by construction and both describe the same run (the background chain runs beside
        the kernel on its own stream, exactly as in the step that is timed)."""
        import torch
        for i in range(warmup):
            self.step(i, mode)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        self.obj_loop.events = []
        t0 = time.perf_counter()
        for i in range(steps):
            self.step(i, mode)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt_ = time.perf_counter() - t0
        ev, self.obj_loop.events = self.obj_loop.events, None
        if dist is not None:
            tt = torch.tensor([dt_], device=self.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt_ = float(tt.item())
        return dt_, sum(e0.elapsed_time(e1) for e0, e1 in ev) / max(1, len(ev))

    # ---- reporting -----------------------------------------------------------------------------------------------
    def rays_per_step(self):
        return self.K_total * self.R if self.wl["scaling"] == "strong" else self.K * self.R * self.world

    def kernel_name(self, mode):
        feat, S, Hd = self.feat, self.S, self.Hd
        if Hd == 32 and S <= 64 and mode != "fp16":
            if mode and S == 64:                         # second-generation kernels (objnerf_bf16v2_body.h)
                return "train_fused_bf16v2f_kernel" if feat else "train_fused_bf16v2_kernel"
            if mode:
                return "train_fused_bf16_kernel<%s, %d>" % ("true" if feat else "false", 64 if S == 64 else 0)
            return "train_fused32_kernel<%s, false, %d>" % ("true" if feat else "false", 64 if S == 64 else 0)
        if Hd == 256 and mode and S in (32, 64, 128):
            if feat:                                       # (on the fused kernels since round 5; no PMC pass recorded for it)
                return "objnerf_train_step, fused hidden-256 path with the feature loss (fwd256_kernel<.., true> + wgrad256_kernel + the head's GEMMs)"
            return "objnerf_train_step, fused hidden-256 path (fwd256_kernel + wgrad256_kernel)"
        return "objnerf_train_step, layer-wise path (batched MFMA GEMMs)"

    def roofline(self, mode, kern_ms, with_peak=False):
        """The dominant launch (objnerf_train_step) against the dense MFMA peak of its operand type: the path is a
        dense contraction with everything else on chip; `traffic` = RECORDED PMC bytes of the same
        launch, `traffic_ratio` against the algorithmic bytes."""
        K, R, S = self.K, self.R, self.S
        fpr = flop_per_ray(S, H=self.Hd, feat=self.feat)
        kname = self.kernel_name(mode)
        achieved = K * R * fpr / (kern_ms * 1e-3) / 1e12
        peak = PEAK_BF16_MFMA_TFLOPS if mode else PEAK_F32_MFMA_TFLOPS
        rec = recorded_counters(kname, K, R, S)
        fused32 = self.Hd == 32 and S <= 64 and mode != "fp16"
        alg = algorithmic_bytes(K, R, S, self.feat, scope="kernel" if fused32 else "step")
        roof = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": rec.get("hbm_bytes_per_launch") if rec else None,
                "traffic_note": ("RECORDED, not measured by this run: " + rec["source"]) if rec else
                                "no recorded PMC pass for this workload",
                "kernel": kname, "kernel_ms": kern_ms, "flop_per_ray": fpr, "algorithmic_bytes_per_launch": alg,
                "algorithmic_bytes_scope": "the fused kernel (what `traffic` was recorded for)" if fused32 else "the step"}
        if self.feat and fused32:
            roof["step_algorithmic_bytes"] = algorithmic_bytes(K, R, S, True, scope="step")
        if rec:
            if rec.get("hbm_bytes_per_launch"):
                roof["traffic_ratio"] = rec["hbm_bytes_per_launch"] / alg
                # the same launch against the HBM roof: recorded bytes / kernel time / sustained bandwidth (the
                # hidden-256 pair moves ~700x its algorithmic bytes: for it THIS is the binding resource, not the MFMA)
                roof["hbm_gbs"] = rec["hbm_bytes_per_launch"] / (kern_ms * 1e-3) / 1e9
                roof["hbm_frac"] = roof["hbm_gbs"] / HBM_SUSTAINED_GBS
                if roof["hbm_frac"] > roof["frac"]:
                    roof["binding"] = "hbm (frac %.2f of %.0f GB/s sustained) -- `frac` is the MFMA figure SURVEY.md 8(d) asks for" % (
                        roof["hbm_frac"], HBM_SUSTAINED_GBS)
            if mode and rec.get("valu_insts_per_launch"):
                floor_ms = rec["valu_insts_per_launch"] * VALU_CYCLES_PER_INST / (VALU_SIMDS * CLOCK_HZ) * 1e3
                roof.update({"valu_insts_per_launch": rec["valu_insts_per_launch"], "valu_floor_ms": floor_ms,
                             "frac_of_valu_floor": floor_ms / kern_ms})
        if with_peak:
            pm = measured_mfma_peak(self.dev, 1 if mode else 0)
            roof["peak_measured"] = pm
            roof["frac_of_measured_peak"] = achieved / pm
        return roof

    def describe(self):
        wl = self.wl
        return {"workload": f"{self.name}: Replica room_0-shaped, {self.K} object MLPs on this GPU"
                            f"{' of %d in total' % self.K_total if wl['scaling'] == 'strong' else ''} (hidden {self.Hd}), "
                            f"{self.R} rays/object/step, {self.S} samples/ray ({self.n1}+{self.n2}), RGB+depth+opacity"
                            f"{'+512-d feature' if self.feat else ''} loss, fused fwd+loss+bwd+AdamW",
                "objects_per_gpu": self.K, "objects_total": self.K_total, "rays_per_object": self.R,
                "samples_per_ray": self.S, "hidden": self.Hd, "feature_head": self.feat,
                "background_mlp": self.bg_loop is not None, "distinct_ray_sets": self.distinct,
                "object_chunks": ((self.K + self.obj_loop.ws.k_chunk - 1) // self.obj_loop.ws.k_chunk
                                  if getattr(self.obj_loop, "ws", None) is not None else 1),
                "chunk_streams": getattr(getattr(self.obj_loop, "ws", None), "lanes", 1),
                "background_rays_on_this_gpu": (self.bg_batches[0]["labels"].shape[1] if self.bg_loop is not None else 0)}

    def free(self):
        import torch
        self.iteration = self.obj_loop = self.bg_loop = self.batches = self.bg_batches = self.arena = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()


OTHER_CONFIGS = [   # (key, base config, overrides, operand modes) -- timed after the headline by the default run
    ("c3", "c3", {}, [False, True]),
    ("c4_share", "c4", {"objects": 15, "scaling": "weak", "bg_ranks": 8}, [False, True]),
    ("c5_share_fp16", "c5", {"bg": False}, ["fp16"]),
    ("c5_share_fp16_feat", "c5", {"bg": False, "feat": True}, ["fp16"]),     # the reference always has the clip branch (model.py:98-101)
]


def other_configs(args, dev):
    """The BASELINE configurations the headline line is not quoted on, each timed by the same procedure (`--other-steps`
    timed steps): c3, configs[3]'s per-GPU share (15 objects with the feature loss) and configs[4]'s per-GPU share
    (64 objects, hidden 256, 8192 x 128) in fp16, without and with the 512-d feature loss (both on the fused hidden-256
    kernels since round 5)."""
    out = {}
    names = {False: "f32", True: "bf16", "fp16": "fp16"}
    for key, base, over, modes in OTHER_CONFIGS:
        wl = dict(CONFIGS[base])
        wl.update(over)
        try:
            w = Workload(key, wl, args, dev, 1, 0, modes[0] if modes[0] == "fp16" else False)
            entry = {"config": w.describe()}
            for mode in modes:
                # hidden-32 entries: 3 x the steps (a step is 1 - 13 ms): the pipelined background chain trails the object
                # chain by a few iterations, which the final synchronize of a 10-step run shows as +0.1 .. 0.2 ms per step
                steps = args.other_steps * (3 if wl["hidden"] == 32 else 1)
                dt, kms = w.timed(mode, steps, 2)
                entry[names[mode]] = {"value": w.rays_per_step() * steps / dt, "unit": "rays/s", "steps": steps,
                                      "ms_per_step": dt / steps * 1e3, "roofline": w.roofline(mode, kms),
                                      "loss_status": int(w.obj_loop.ws.status.item())}
            w.free()
            out[key] = entry
        except Exception as e:      # a failing extra must not cost the headline line; it is reported, not hidden
            out[key] = {"error": f"{type(e).__name__}: {e}"}
    return out


def share_curve(args, dev, steps):
    """Emulated per-rank shares on ONE GPU (hardware-free; NO scaling curve is claimed): this process stands for one of
    N = 1, 2, 4, 8 ranks -- configs[3] (strong scaling: 120 objects with the feature loss dealt to the ranks, 120 / N
    here) and configs[1] (weak: 50 objects per rank) -- and takes that rank's 1200 / N of the iteration's background
    rays as dist.shard_rays deals them.  NOT in it: the latency of the iteration's collectives (the background
    gradient's 729 KB all-reduce, the pre-step int32[4]) and any xGMI effect.  efficiency = per-rank throughput at N
    over the N = 1 figure (weak), or the speed-up of the step over N (strong)."""
    names = {False: "f32", True: "bf16"}
    out = {"note": "one GPU standing for one of N ranks; collective latency NOT included; no measured multi-GPU curve",
           "steps": steps, "rows": []}
    for cfgname, base in (("c4", "c4"), ("c2", "c2")):
        t1 = {}
        for n in (1, 2, 4, 8):
            wl = dict(CONFIGS[base])
            if wl["scaling"] == "strong":
                wl.update(objects=wl["objects"] // n, scaling="weak")
            wl["bg_ranks"] = n
            try:
                w = Workload(f"{cfgname}_share_of_{n}", wl, args, dev, 1, 0, False)
                for mode in (False, True):
                    dt, kms = w.timed(mode, steps, 3)
                    ms = dt / steps * 1e3
                    if n == 1:
                        t1[mode] = ms
                    strong = CONFIGS[base]["scaling"] == "strong"
                    eff = (t1[mode] / ms / n) if strong else (t1[mode] / ms)
                    out["rows"].append({"config": cfgname, "ranks": n, "dtype": names[mode], "objects_on_this_rank": w.K,
                                        "background_rays_on_this_rank": int(w.bg_batches[0]["labels"].shape[1]),
                                        "ms_per_step": ms, "kernel_ms": kms, "rays_per_s_per_rank": w.K * w.R / (ms * 1e-3),
                                        "scaling": CONFIGS[base]["scaling"], "efficiency_vs_1": eff})
                w.free()
            except Exception as e:
                out["rows"].append({"config": cfgname, "ranks": n, "error": f"{type(e).__name__}: {e}"})
    return out


def native_frame(dev, bf16=False, n_objects=50, frames=4):
    """The mapping loop at the reference's NATIVE shape on the driver's clock (north_star's "stratified ray sampling"
    included): synthetic 1200 x 680 frames, 50 objects + background, room_0 hyper-parameters (100 iterations of 120 rays x
    10 samples per object, background 1200 rays x 14 samples; mapping.IncrementalMapper): per frame the ingestion, the
    sample pools (objnerf_sample_rays_stacked, seeded draws), the 100 iterations with the copy-back.  Medians over the
    frames after the first (which pays the allocations)."""
    import numpy as np
    import torch
    from openobj_amd import cfg as ocfg, mapping, synthetic
    c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev), **{"trainer.part_mode": 0}))
    m = mapping.IncrementalMapper(c, bf16=bf16)
    sync = torch.cuda.synchronize
    rows = []
    for i in range(frames):
        s = synthetic.grid_frame(i, n_objects)
        sync(); t0 = time.perf_counter()
        m.ingest(s, i)
        sync(); t1 = time.perf_counter()
        m._ensure_stack()
        pool, bg_pool = m.sample_pools()
        sync(); t2 = time.perf_counter()
        real = m.sample_pools
        m.sample_pools = lambda: (pool, bg_pool)
        m.train_frame()
        m.sample_pools = real
        sync(); t3 = time.perf_counter()
        rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
    r = np.median(np.array(rows[1:]), axis=0)
    n_it = c.n_iter_per_frame
    rays = n_objects * c.n_per_optim * n_it
    out = {"workload": f"{n_objects} objects + background, 1200 x 680 frames, {n_it} iterations x {c.n_per_optim} rays x "
                       f"{c.n_bins_cam2surface + c.n_bins} samples per object, background {c.n_per_optim_bg} rays x "
                       f"{c.n_bins_cam2surface_bg + c.n_bins} samples, hidden 32 / 128, {'bf16' if bf16 else 'fp32'}",
           "ingest_ms": float(r[0]), "sample_pools_ms": float(r[1]), "iterations_ms": float(r[2]),
           "ms_per_iteration": float(r[2] / n_it), "frame_ms": float(r.sum()),
           "object_rays_per_s": rays / (float(r.sum()) * 1e-3), "frames_timed": frames - 1,
           "hbm_gb": torch.cuda.memory_allocated() / 2 ** 30}
    del m
    sync()
    torch.cuda.empty_cache()
    return out


# ----------------------------------------------------------------------------------------------------------------------
# What stdout carries.  Round 5's line had grown to 20.8 KB and the driver stopped parsing it (BENCH_r05.parsed = null):
# the line is now a fixed, flat selection (< LINE_LIMIT bytes, asserted) and the full report goes to --detail-out.
# ----------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 4096
LINE_TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                 "vs_baseline", "dtype", "data", "dry_launch", "rays_per_sec_per_gpu", "psnr_delta_db", "psnr_delta_ci95_db",
                 "metric_version")
LINE_CONFIG_KEYS = ("workload", "objects_per_gpu", "objects_total", "rays_per_object", "samples_per_ray", "hidden",
                    "feature_head", "background_mlp", "parallelism", "collectives_per_step", "rccl_ranks", "objects_per_rank",
                    "launched_by", "loss_status", "backend", "collectives_ok")
LINE_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms", "flop_per_ray",
                      "algorithmic_bytes_per_launch", "traffic_ratio", "traffic_source", "peak_measured", "frac_of_measured_peak")
LINE_CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _r6(x):
    """Numbers of the line to 6 significant digits (the detail file keeps full precision)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    return float("%.6g" % x)


def _clip(sx, n):
    return sx if len(sx) <= n else sx[:n - 3] + "..."


def short_line(out: dict, detail_path=None) -> str:
    """The ONE stdout line: the contract's keys, `roofline` and `cpu_baseline` as flat objects, the flat `summary`
    numbers, and the path of the detail file -- nothing nested deeper, no per-config blocks, no prose beyond the workload
    name.  Always < LINE_LIMIT bytes: an oversized line is an error here, not a surprise in the driver's parser."""
    line = {k: _r6(out[k]) for k in LINE_TOP_KEYS if k in out}
    cfg = out.get("config", {})
    line["config"] = {k: (_clip(cfg[k], 200) if isinstance(cfg[k], str) else cfg[k]) for k in LINE_CONFIG_KEYS if k in cfg}
    if len(line["config"].get("objects_per_rank", [])) > 16:
        line["config"].pop("objects_per_rank")
    if "roofline" in out:
        rf = dict(out["roofline"])
        if rf.get("traffic") is not None and "traffic_source" not in rf:
            rf["traffic_source"] = "recorded rocprofv3 --pmc pass of the same launch (profiles/), not read by this run"
        line["roofline"] = {k: (_r6(rf[k]) if not isinstance(rf[k], str) else _clip(rf[k], 120))
                            for k in LINE_ROOFLINE_KEYS if k in rf}
    if "cpu_baseline" in out:
        line["cpu_baseline"] = {k: (_r6(out["cpu_baseline"][k]) if not isinstance(out["cpu_baseline"][k], str)
                                    else _clip(out["cpu_baseline"][k], 200)) for k in LINE_CPU_KEYS if k in out["cpu_baseline"]}
    if "dist_selftest" in out:
        line["dist_selftest"] = {k: _r6(v) for k, v in out["dist_selftest"].items() if not isinstance(v, (dict, list, str))}
    if detail_path:
        line["detail"] = detail_path
    if "summary" in out:                                    # LAST (the driver records the tail): flat numbers only
        line["summary"] = {k: _r6(v) for k, v in out["summary"].items() if isinstance(v, (int, float)) or v is None}
    text = json.dumps(line)
    if len(text) >= LINE_LIMIT:                             # shed the optional parts before giving up
        for k in ("summary", "dist_selftest"):
            line.pop(k, None)
            text = json.dumps(line)
            if len(text) < LINE_LIMIT:
                break
    if len(text) >= LINE_LIMIT:
        raise RuntimeError(f"bench.py: the stdout line is {len(text)} bytes (limit {LINE_LIMIT})")
    return text


def emit(out: dict, json_fd: int, detail_path=None):
    """Full report -> the detail file (best effort: an unwritable path must not cost the line); short line -> stdout."""
    written = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1)
            written = detail_path
        except OSError as e:
            sys.stderr.write(f"bench.py: detail file {detail_path} not written: {e}\n")
    os.write(json_fd, (short_line(out, written) + "\n").encode())



def collective_times(dev, dist, n=50):
    """The iteration's two collectives ALONE, on the device, every rank together (RCCL; a single rank under
    OBJNERF_DIST_SELFTEST=1 measures the backend's fixed cost per call): collective 1 = the pre-step int32[4] SUM,
    collective 2 = the replicated background network's gradient, fp32 SUM of 182 339 + 4 floats (729 KB).  ms per call,
    HIP events around `n` back-to-back calls after 5 warm-up calls; max over ranks.  The step hides collective 2 under
    the object kernel (openobj_amd.train.ShardedIteration): these are the numbers that claim is to be checked against."""
    import torch
    from openobj_amd import dist as odist
    pre = torch.zeros(4, dtype=torch.int32, device=dev)
    flat = torch.zeros(BG_GRAD_FLOATS, device=dev)
    out = {}
    for name, t in (("collective1_ms", pre), ("collective2_ms", flat)):
        for _ in range(5):
            odist.allreduce_sum_(t)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            odist.allreduce_sum_(t)
        e1.record()
        torch.cuda.synchronize()
        tt = torch.tensor([e0.elapsed_time(e1) / n], device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        out[name] = float(tt.item())
    out["collective2_bytes"] = BG_GRAD_FLOATS * 4
    out["ranks"] = dist.get_world_size()
    return out


def dry_launch(args, world, rank, json_fd):
    """The N-rank launch on CPU: gloo, the iteration's two collectives on host buffers of the real sizes, the bench's
    barrier / max-over-ranks timing; no kernels (value 0)."""
    import torch
    import torch.distributed as dist
    from openobj_amd import dist as odist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if os.environ.get("OBJNERF_BENCH_DIE_EARLY") == str(rank):     # test hook: a rank that dies before the rendezvous
        return 4
    dist.init_process_group("gloo")
    wl = dict(CONFIGS[args.config or "c2"])
    if wl["scaling"] == "strong":
        lo, hi = odist.shard_objects(wl["objects"], world, rank)
        K = hi - lo
    else:
        K = wl["objects"]
    per_rank = torch.zeros(world, dtype=torch.int64)
    per_rank[rank] = K
    dist.all_reduce(per_rank)
    flat = torch.zeros(BG_GRAD_FLOATS)
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pre = odist.pack_pre(torch.tensor([0, 1 if rank == 0 else 0], dtype=torch.int32),
                             torch.tensor([[600, 500]], dtype=torch.int32), "cpu")
        odist.allreduce_sum_(pre)                                  # collective 1
        gflags, bg_counts, _ = odist.unpack_pre(pre)
        flat.fill_(1.0)
        work = odist.allreduce_sum_async(flat)                     # collective 2
        work.wait()
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0])
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ok = gflags.tolist() == [0, 1] and bg_counts.tolist() == [[600 * world, 500 * world]] and float(flat[0]) == world
    if rank == 0:
        out = {"metric": "training rays/sec/GPU @64 samples/ray, 50 obj; PSNR delta vs ref", "value": 0.0,
               "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": float(tt.item()) / max(1, args.steps) * 1e3, "higher_is_better": True,
               "scaling": wl["scaling"], "vs_baseline": None, "dtype": args.dtype, "data": "none", "dry_launch": True,
               "config": {"workload": "dry launch: collectives only (gloo, host buffers), no kernels",
                          "rccl_ranks": dist.get_world_size(), "backend": "gloo",
                          "objects_per_rank": per_rank.tolist(), "collectives_per_step": 2,
                          "collectives_ok": bool(ok)}}
        emit(out, json_fd, args.detail_out)        # (no detail file unless asked for: the dry launch has nothing more to say)
    dist.destroy_process_group()
    if os.environ.get("OBJNERF_BENCH_FAIL_RANK") == str(rank):     # test hook: a rank that fails after the run
        return 3
    return 0 if ok else 1


def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))

    # stdout carries exactly ONE line (the JSON, rank 0).  Libraries print there too (RCCL's version banner at the
    # first collective): file descriptor 1 is pointed at stderr for the run and the line goes to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.dry_launch:
        sys.exit(dry_launch(args, world, rank, json_fd))

    import torch
    cfgname = args.config or "c2"
    wl = dict(CONFIGS[cfgname])
    for k_arg, k_wl in (("objects", "objects"), ("rays", "rays"), ("n_cam2surf", "n1"), ("n_bins", "n2"),
                        ("hidden", "hidden"), ("feat", "feat"), ("bg_ranks", "bg_ranks")):
        if getattr(args, k_arg) is not None:
            wl[k_wl] = getattr(args, k_arg)
    default_line = (args.config is None and args.dtype == "f32" and args.bg and
                    all(getattr(args, k) is None for k in ("objects", "rays", "n_cam2surf", "n_bins", "hidden", "feat")))

    dist = None
    use_dist = world > 1 or (os.environ.get("OBJNERF_DIST_SELFTEST") == "1" and "RANK" in os.environ)
    if local_rank >= torch.cuda.device_count():
        sys.stderr.write(f"bench.py: rank {rank} wants cuda:{local_rank} but the node shows "
                         f"{torch.cuda.device_count()} GPU(s)\n")
        sys.exit(2)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    mode = {"f32": False, "bf16": True, "fp16": "fp16"}[args.dtype]      # ops.precision_bits
    w = Workload(cfgname, wl, args, dev, world, rank, mode)
    dt, kern_ms = w.timed(mode, args.steps, args.warmup, dist if use_dist else None)
    status = int(w.obj_loop.ws.status.item())
    # the opt-in bf16-operand mode beside the fp32 headline (same step, same batches)
    bf16_extra = None
    if not mode and args.bf16_line:
        bf16_extra = w.timed(True, args.steps, args.warmup, dist if use_dist else None)
    per_rank = [w.K]
    coll = None
    if use_dist:
        pr = torch.zeros(world, dtype=torch.int64, device=dev)
        pr[rank] = w.K
        dist.all_reduce(pr)
        per_rank = pr.tolist()
        coll = collective_times(dev, dist)

    if rank == 0:
        rays_per_step = w.rays_per_step()
        value = rays_per_step * args.steps / dt
        cfg_block = w.describe()
        cfg_block.update({"parallelism": f"objects sharded x{world}", "collectives_per_step": 2 if use_dist else 0,
                          "rccl_ranks": dist.get_world_size() if use_dist else 1, "objects_per_rank": per_rank,
                          "launched_by": "bench.py --gpus" if os.environ.get("OBJNERF_BENCH_LAUNCHED") else
                                         ("torchrun" if "RANK" in os.environ else "single process"),
                          "loss_status": status})
        out = {
            "metric": "training rays/sec/GPU @64 samples/ray, 50 obj; PSNR delta vs ref",
            "value": value, "unit": "rays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": wl["scaling"],
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": cfg_block,
            "rays_per_sec_per_gpu": value / world,
            "roofline": w.roofline(mode, kern_ms, with_peak=not args.no_peak),
        }
        if coll is not None:
            out["dist_selftest"] = coll
        if bf16_extra is not None:
            bdt, bk = bf16_extra
            out["bf16_mode"] = {"value": rays_per_step * args.steps / bdt, "unit": "rays/s",
                                "ms_per_step": bdt / args.steps * 1e3,
                                "roofline": w.roofline(True, bk, with_peak=not args.no_peak),
                                "note": "OBJNERF_TRAIN_BF16: bf16 MFMA operands, fp32 accumulate / master weights / "
                                        "compositing / AdamW; PSNR-gated (tests/test_bf16_gpu.py), not 1e-4 parity"}
    Hd, feat = w.Hd, w.feat
    w.free()
    if rank == 0:
        if world == 1 and default_line and not args.no_other_configs:
            out["other_configs"] = other_configs(args, dev)
            # (the emulated 1 / 2 / 4 / 8-rank share table is NOT part of the default run: --share-curve-out)
            for key, fn in (("native_frame", lambda: native_frame(dev)), ("native_frame_bf16", lambda: native_frame(dev, True))):
                try:
                    out["other_configs"][key] = fn()
                except Exception as e:
                    out["other_configs"][key] = {"error": f"{type(e).__name__}: {e}"}
        if world == 1 and args.share_curve_out:
            sc = share_curve(args, dev, args.steps)
            with open(args.share_curve_out, "w") as f:
                json.dump(sc, f, indent=1)
            out["share_curve_file"] = args.share_curve_out
        if world == 1 and not args.no_psnr and Hd == 32:
            ps = psnr_block(dev, args.psnr_seeds, with_bf16=args.bf16_line or mode is True, with_fp16=mode == "fp16")
            if ps is not None:
                out["psnr"] = ps
                key = args.dtype
                # the metric's "PSNR delta vs ref" is that of the TRAINED model: the 300-iteration ensemble difference
                # (with its 95 % half-width) is the headline, the 50-iteration per-seed mean a secondary figure
                out["psnr_delta_db"] = ps["iter300"][key]["delta_db"]
                out["psnr_delta_ci95_db"] = ps["iter300"][key]["ci95_db"]
                out["psnr_delta_iter50_db"] = ps["iter50"][key]["mean_delta_db"]
                out["psnr_delta_iter50_ci95_db"] = ps["iter50"][key]["ci95_db"]
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(feat)
        # what the line's figures mean has changed before (round 4: psnr_delta_db became the 300-iteration ensemble figure,
        # kernel_ms moved inside the timed steps); versioned since round 5, whose changes are listed here
        out["metric_version"] = 6
        out["metric_changes"] = ("v6: stdout carries a fixed short line (< 4 KB: contract keys, flat roofline / cpu_baseline / "
                                 "summary); this full report is the --detail-out file; the emulated share table left the "
                                 "default run (--share-curve-out).  v5: objnerf_train_step applies AdamW in its last launch, so roofline.kernel_ms (HIP events "
                                 "around objnerf_train_step, inside the timed steps) now includes the optimiser; the "
                                 "background chain of step i may run under the object kernel of step i + 1 (--no-pipeline "
                                 "restores the per-step join); algorithmic_bytes_per_launch of the feature configs is the "
                                 "fused kernel's scope; c5 trains 64 distinct ray sets (c5_share_fp16_feat: the same with the 512-d "
                                 "feature loss, on the fused hidden-256 kernels); other_configs.c4_share stands "
                                 "for one of 8 ranks WITHOUT the gradient all-reduce it would take part in")
        # LAST in the line (the driver records its tail): the figures a reader looks for first
        summ = {"f32_rays_per_s": out["value"] if args.dtype == "f32" else None, "f32_ms_per_step": out["ms_per_step"]
                if args.dtype == "f32" else None, "f32_roofline_frac": out["roofline"]["frac"] if args.dtype == "f32" else None}
        if "bf16_mode" in out:
            summ.update({"bf16_rays_per_s": out["bf16_mode"]["value"], "bf16_ms_per_step": out["bf16_mode"]["ms_per_step"],
                         "bf16_roofline_frac": out["bf16_mode"]["roofline"]["frac"],
                         "bf16_kernel_ms": out["bf16_mode"]["roofline"]["kernel_ms"]})
        oc = out.get("other_configs", {})
        for key in ("c3", "c4_share", "c5_share_fp16", "c5_share_fp16_feat"):
            for m_ in ("f32", "bf16", "fp16"):
                if isinstance(oc.get(key), dict) and m_ in oc[key]:
                    summ[f"{key}_{m_}_rays_per_s"] = oc[key][m_]["value"]
                    summ[f"{key}_{m_}_ms_per_step"] = oc[key][m_]["ms_per_step"]
        for key in ("native_frame", "native_frame_bf16"):
            if isinstance(oc.get(key), dict) and "frame_ms" in oc[key]:
                summ[f"{key}_ms"] = oc[key]["frame_ms"]
                summ[f"{key}_ms_per_iteration"] = oc[key]["ms_per_iteration"]
        if isinstance(oc.get("share_curve"), dict):
            summ["share_curve_efficiency_at_8"] = {f"{r['config']}_{r['dtype']}": round(r["efficiency_vs_1"], 3)
                                                   for r in oc["share_curve"]["rows"] if r.get("ranks") == 8 and "dtype" in r}
        for k_ in ("psnr_delta_db", "psnr_delta_ci95_db"):
            if k_ in out:
                summ[k_] = out[k_]
        if "cpu_baseline" in out:
            summ["cpu_baseline_rays_per_s"] = out["cpu_baseline"]["value"]
        out["summary"] = summ
        sys.stdout.flush()
        emit(out, json_fd, args.detail_out or os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
