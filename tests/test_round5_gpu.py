"""GPU: round 5 -- the optimiser inside objnerf_train_step (objnerf_train_args.optim, ABI 7), the step's own label
statistics (OBJNERF_TRAIN_SELF_COUNTS), the status word's non-finite bit, the pipelined iteration."""
import numpy as np
import pytest
import torch

from conftest import T
from openobj_amd import cfg as ocfg
from openobj_amd import init as obj_init
from openobj_amd import ops, optim, synthetic, trainer
from openobj_amd import train as otrain

pytestmark = pytest.mark.gpu
KEYS = ["pts", "z", "gt_depth", "gt_rgb", "labels"]


def _arena(K, H, dev, seed):
    st = obj_init.init_stacked(K, H, 512, seed=seed)
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    arena.load_stacked([q.clone() for q in st])
    arena.scale.fill_(2.0 if H == 32 else 5.0)
    return arena


def _batch(K, R, n1, n2, seed, dev, feat=False, empty_obj=None):
    b = synthetic.random_batch(K, R, n1, n2, seed=seed, feat_dim=512 if feat else 0)
    if empty_obj is not None:
        b["labels"][empty_obj][b["labels"][empty_obj] == 1] = 0       # render_rays.py:89-94: the early return for ALL objects
    return {k: T(b[k]).to(dev) for k in KEYS + (["gt_feat"] if feat else [])}


@pytest.mark.parametrize("case", ["obj32", "obj32_bf16", "obj32_feat", "obj32_bf16_feat", "bg128", "bg128_bf16", "bg128_feat",
                                  "bg128_k3", "bg128_k3_feat", "bg128_bf16_feat"])
def test_optimiser_inside_the_step_equals_the_separate_launch(dev, case):
    """objnerf_train_step(optim=) -- AdamW applied by the launch that reduces the partial gradients (finalize_kernel for
    the fused hidden-32 kernels, reduce_parts_kernel for the one-launch hidden-128 iteration, a trailing launch
    elsewhere) -- leaves parameters, both moments, the per-group step counters and the gradient BIT-IDENTICAL to
    objnerf_train_step followed by objnerf_adamw_step_flags, over four iterations of which the second hits the
    cross-object early return (groups skipped like .grad = None)."""
    H = 32 if case.startswith("obj") else 128
    K = 3 if (case.startswith("obj") or "k3" in case) else 1
    feat = "feat" in case
    bf16 = "bf16" in case
    R, n1, n2 = (96, 16, 48) if H == 32 else (150, 5, 9)
    runs = []
    for fused in (False, True):
        arena = _arena(K, H, dev, seed=5)
        opt = optim.ArenaAdamW(arena, lr=1e-3, weight_decay=0.013)
        ws = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, precision=bf16)
        mask = arena.has_grad_mask(feat)
        for it in range(4):
            b = _batch(K, R, n1, n2, 300 + it, dev, feat, empty_obj=(K - 1) if it == 1 else None)
            if fused:
                ops.train_step(arena, ws, b, with_feat=feat, bf16=bf16, optim=opt)
            else:
                ops.train_step(arena, ws, b, with_feat=feat, bf16=bf16)
                opt.step(ws.grads, mask, flags=ws.flags)
        torch.cuda.synchronize()
        runs.append((arena.params.clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone(), opt.group_steps.clone(),
                     ws.grads.clone(), ws.loss_terms.clone(), ws.flags.clone(), int(ws.status.item())))
    a, b = runs
    assert a[3].tolist() == b[3].tolist()
    for x, y, name in zip(a[:3], b[:3], ["params", "exp_avg", "exp_avg_sq"]):
        assert torch.equal(x, y), (case, name, float((x - y).abs().max()))
    assert torch.equal(a[4], b[4]) and torch.equal(a[5], b[5]) and torch.equal(a[6], b[6]) and a[7] == b[7] == 0
    assert a[3][0].item() == 4 and a[3][1].item() == 3          # trunk stepped 4 times, the colour branch 3


@pytest.mark.parametrize("H,K,R,n1,n2", [(32, 5, 64, 16, 48), (128, 1, 150, 5, 9), (128, 3, 40, 5, 9), (128, 1, 64, 32, 96)])
def test_step_derives_the_label_statistics_itself(dev, H, K, R, n1, n2):
    """OBJNERF_TRAIN_SELF_COUNTS (no global_flags / global_counts): counts and flags come out as objnerf_label_counts
    computes them and the step equals the one that was handed them -- for the fused hidden-32 kernel (flags derived by
    every workgroup from the K counts), the one-launch hidden-128 iteration (K = 1: counted inside the kernel) and the
    layer-wise chain (long rays)."""
    for empty in (None, K - 1):
        arena = _arena(K, H, dev, seed=9)
        b = _batch(K, R, n1, n2, 77, dev, empty_obj=empty)
        counts, flags = ops.label_counts(b["labels"])
        ws1 = ops.TrainWorkspace(arena, K, R, n1 + n2, False)
        ops.train_step(arena, ws1, b, global_flags=flags, global_counts=counts)
        ws2 = ops.TrainWorkspace(arena, K, R, n1 + n2, False)
        ws2.flags.fill_(7)
        ws2.counts.fill_(-1)
        ops.train_step(arena, ws2, b)
        torch.cuda.synchronize()
        assert torch.equal(ws2.counts, counts) and torch.equal(ws2.flags, flags), (ws2.counts, counts, ws2.flags, flags)
        assert torch.equal(ws1.grads, ws2.grads) and torch.equal(ws1.loss_terms, ws2.loss_terms)
        assert flags.tolist() == ([1, 0] if empty is not None else [0, 0])


@pytest.mark.parametrize("path", ["fused32", "small128", "layerwise128", "layerwise32", "fused256_fp16"])
def test_status_reports_a_non_finite_loss(dev, path):
    """The status word, the same on EVERY path (round 5 set the second bit on the hidden-32 and small hidden-128 paths
    only): bit 0 = a per-object term above 1e5 (render_rays.py:109-111, the reference exits), bit 1 = a term that is not
    finite.  A NaN in the density head reaches every ray's weights: the reference's loss is NaN too (and it carries on --
    `nan > 100000` is False); here the step says so and the host contract (render_rays.check_status) is a warning, not
    LossExplode.  (A NaN that only reaches a ReLU is swallowed by it in every build -- fmaxf returns the other operand --
    where torch's relu propagates it: DESIGN.md section 2.)"""
    import warnings
    from openobj_amd.render_rays import LossExplode, check_status
    H = {"fused32": 32, "layerwise32": 32, "small128": 128, "layerwise128": 128, "fused256_fp16": 256}[path]
    K, R, n1, n2 = {32: (2, 32, 16, 48), 128: (1, 40, 5, 9), 256: (2, 64, 16, 48)}[H]
    lw = path.startswith("layerwise")
    prec = "fp16" if path == "fused256_fp16" else False
    arena = _arena(K, H, dev, seed=3)
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False, layerwise=lw, precision=prec) if (lw or prec) else \
        ops.TrainWorkspace(arena, K, R, n1 + n2, False)
    b = _batch(K, R, n1, n2, 5, dev)
    ops.train_step(arena, ws, b, layerwise=lw, bf16=prec)
    assert int(ws.status.item()) == 0 and bool(torch.isfinite(ws.loss_terms).all())
    assert check_status(ws.status) == 0
    if H == 256:      # (the fused hidden-256 kernels clamp the density head's argument: v_max / v_min swallow a NaN weight like a ReLU)
        b["gt_depth"][K - 1, 0] = float("nan")             # a NaN target of the last object: its depth term is NaN in the reference too
    else:
        arena.views()[8][K - 1, 0, 3] = float("nan")       # alpha_linear.weight of the last object
    ops.train_step(arena, ws, b, layerwise=lw, bf16=prec)
    torch.cuda.synchronize()
    assert int(ws.status.item()) & 2, int(ws.status.item())
    assert not bool(torch.isfinite(ws.loss_terms[K - 1]).all())
    if K > 1:
        assert bool(torch.isfinite(ws.loss_terms[0]).all())
    with warnings.catch_warnings(record=True) as rec:        # the host contract: carry on, loudly
        warnings.simplefilter("always")
        assert check_status(ws.status) & 2
    assert any(issubclass(w.category, RuntimeWarning) for w in rec)
    with pytest.raises(LossExplode):
        check_status(torch.tensor([3]))


def test_pipelined_iteration_equals_the_joined_one(dev):
    """ShardedIteration(pipelined=True): the caller's stream does not wait for the background chain at the end of a step
    (bench.py, resident batches).  After join() parameters, moments and every iteration's loss terms equal the
    step-by-step order bit for bit."""
    def make_cfg():
        c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev)))
        c.obj_id = 1
        return c

    def run(pipelined):
        torch.manual_seed(21)
        ts = [trainer.Trainer(make_cfg()) for _ in range(3)]
        c = make_cfg()
        c.hidden_feature_size, c.obj_scale, c.obj_id = 128, 5.0, 0
        torch.manual_seed(22)
        bg = trainer.Trainer(c)
        obj_loop = otrain.HipTrainLoop(make_cfg(), ts, with_feat=False)
        bg_loop = otrain.BackgroundLoop(c, bg)
        it = otrain.ShardedIteration(obj_loop, bg_loop, overlap=True, resident=True, pipelined=pipelined, device=dev)
        batches = [(_batch(3, 96, 16, 48, 30 + i, dev), _batch(1, 200, 5, 9, 40 + i, dev)) for i in range(5)]
        ot = torch.zeros(5, 3, 4, device=dev)
        torch.cuda.synchronize()
        bts = []
        for i, (ob, bgb) in enumerate(batches):
            o, bt = it.step(ob, bgb)
            ot[i].copy_(o)
            bts.append(bt)
        it.join()
        last_bg = bts[-1].clone()
        torch.cuda.synchronize()
        return obj_loop.arena.params.clone(), bg.arena.params.clone(), bg_loop.opt.exp_avg.clone(), ot, last_bg
    a, b = run(False), run(True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


@pytest.mark.parametrize("C", [512, 64, 40])
@pytest.mark.parametrize("with_opt", [False, True])
def test_feature_step_at_other_feature_widths(dev, C, with_opt):
    """The feature-loss step of the fused hidden-32 kernels at clip_point_feature_size = 512 (round 6's route: Gram matrix,
    label counts and the [W_of | b_of] copy inside feat_pre_kernel, the head's finish inside finalize_kernel), 64 (feat_pre_kernel
    + the two split-K GEMMs + feat_finish_kernel: the route for C != 512) and 40 (not a multiple of 16: batched GEMM +
    feat_rowstats_kernel ahead of the fused kernel) against the oracle -- the reference's op sequence -- on the same batch: the
    loss terms and all 19 gradients; with an optimiser attached, AdamW inside the step equals the separate launch bit for bit.
    (No fixture has a feature width other than 512; cfg.clip_point_feature_size is a reference option, cfg.py.)"""
    from oracle import objnerf_oracle as O
    K, R, n1, n2 = 2, 40, 8, 24
    st = obj_init.init_stacked(K, 32, C, seed=11)
    b = synthetic.random_batch(K, R, n1, n2, seed=91, feat_dim=C)
    batch = {k: T(b[k]).to(dev) for k in KEYS + ["gt_feat"]}
    arena = ops.ParamArena(K, ops.NetShape(32, C, 6), dev)
    arena.load_stacked([q.clone() for q in st])
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
    opt = optim.ArenaAdamW(arena, lr=1e-3, weight_decay=0.013) if with_opt else None
    ops.train_step(arena, ws, batch, with_feat=True, optim=opt)
    torch.cuda.synchronize()
    assert int(ws.status.item()) == 0
    fcr = [p.clone().requires_grad_(True) for p in st[:18]]
    Br = st[18].clone().requires_grad_(True)
    loss, terms = O.train_forward_loss(fcr, Br, torch.full((K,), 2.0), *[T(b[k]) for k in ["pts", "gt_depth", "gt_rgb", "labels", "z"]],
                                       gt_feat=T(b["gt_feat"]), return_terms=True)
    grads = torch.autograd.grad(loss, fcr + [Br])
    t = ws.loss_terms.cpu()
    total = (t[:, 0] + 5 * t[:, 1] + 10 * t[:, 2] + 5 * t[:, 3]).sum().item()
    assert abs(total - loss.item()) < 1e-4 * abs(loss.item()), (total, loss.item())
    gv = arena.views(ws.grads)
    for i, gr in enumerate(grads):
        err = (gv[i].cpu() - gr).abs().max().item()
        assert err < 2e-4 * max(1e-3, gr.abs().max().item()), (C, i, ops.TENSOR_NAMES[i], err)
    if with_opt:            # the same step with the optimiser as a separate launch: parameters and moments bit-equal
        arena2 = ops.ParamArena(K, ops.NetShape(32, C, 6), dev)
        arena2.load_stacked([q.clone() for q in st])
        ws2 = ops.TrainWorkspace(arena2, K, R, n1 + n2, True)
        opt2 = optim.ArenaAdamW(arena2, lr=1e-3, weight_decay=0.013)
        ops.train_step(arena2, ws2, batch, with_feat=True)
        opt2.step(ws2.grads, arena2.has_grad_mask(True), flags=ws2.flags)
        torch.cuda.synchronize()
        assert torch.equal(ws.grads, ws2.grads)
        assert torch.equal(arena.params, arena2.params) and torch.equal(opt.exp_avg, opt2.exp_avg)
        assert torch.equal(opt.exp_avg_sq, opt2.exp_avg_sq)
