"""GPU: the opt-in bf16-operand mode of the fused training kernel (OBJNERF_TRAIN_BF16).

The reference computes in fp32 (train.py:74, AMP off), so this mode is NOT held to the 1e-4 parity bar; it is
held to (a) gradients / losses that agree with the fp32 kernel to bf16 rounding noise and (b) the same
reconstruction quality (PSNR) as the reference on the integration fixture."""
import numpy as np
import pytest
import torch

from conftest import T
from openobj_amd import ops, synthetic
from test_api_gpu import _train_and_psnr

pytestmark = pytest.mark.gpu


def _arena(golden, K, dev):
    g = golden("g9_psnr_nofeat")
    fc = [T(g[f"fc0_{i}"])[:1].repeat(K, *([1] * (T(g[f"fc0_{i}"]).dim() - 1))).clone() for i in range(18)]
    gen = torch.Generator().manual_seed(K)
    fc = [p + 0.05 * torch.randn(p.shape, generator=gen) * p.abs().mean() for p in fc]
    from oracle import objnerf_oracle as O
    B = O.icosa_dirs()[None].repeat(K, 1, 1) + 0.01 * torch.randn(K, 21, 3, generator=gen)
    arena = ops.ParamArena(K, ops.NetShape(), dev)
    arena.load_stacked(fc + [B])
    return arena


@pytest.mark.parametrize("shape", [(3, 300, 16, 48), (2, 211, 5, 9), (2, 128, 8, 24)])
def test_bf16_step_close_to_fp32(golden, dev, shape):
    K, R, n1, n2 = shape
    arena = _arena(golden, K, dev)
    b = synthetic.random_batch(K, R, n1, n2, seed=31 + R)
    batch = {k: T(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
    ws32 = ops.TrainWorkspace(arena, K, R, n1 + n2, False)
    ws16 = ops.TrainWorkspace(arena, K, R, n1 + n2, False)
    ops.train_step(arena, ws32, batch)
    ops.train_step(arena, ws16, batch, bf16=True)
    torch.cuda.synchronize()
    assert int(ws16.status.item()) == 0
    t32, t16 = ws32.loss_terms.cpu(), ws16.loss_terms.cpu()
    np.testing.assert_allclose(t16[:, :3], t32[:, :3], rtol=2e-2, atol=2e-3)
    g32, g16 = arena.views(ws32.grads), arena.views(ws16.grads)
    for i in list(range(14)) + [18]:
        a, r = g16[i].double().cpu(), g32[i].double().cpu()
        rel = float((a - r).norm() / (r.norm() + 1e-12))
        print(ops.TENSOR_NAMES[i], "rel err", round(rel, 4))
        assert rel < 0.15, (i, ops.TENSOR_NAMES[i], rel)
    for i in ops.FEAT_TENSORS:
        assert float(g16[i].abs().max()) == 0.0


@pytest.mark.parametrize("shape", [(3, 300, 16, 48), (2, 211, 5, 9)])
def test_bf16_feature_step_close_to_fp32(golden, dev, shape):
    """The bf16 mode with the 512-d feature-distillation loss (BASELINE configs[2]) against the fp32 kernel."""
    K, R, n1, n2 = shape
    arena = _arena(golden, K, dev)
    b = synthetic.random_batch(K, R, n1, n2, seed=41 + R, feat_dim=512)
    batch = {k: T(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels", "gt_feat"]}
    ws32 = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
    ws16 = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
    ops.train_step(arena, ws32, batch, with_feat=True)
    ops.train_step(arena, ws16, batch, with_feat=True, bf16=True)
    torch.cuda.synchronize()
    assert int(ws16.status.item()) == 0
    np.testing.assert_allclose(ws16.loss_terms.cpu(), ws32.loss_terms.cpu(), rtol=2e-2, atol=2e-3)
    g32, g16 = arena.views(ws32.grads), arena.views(ws16.grads)
    for i in range(19):
        a, r = g16[i].double().cpu(), g32[i].double().cpu()
        rel = float((a - r).norm() / (r.norm() + 1e-12))
        print(ops.TENSOR_NAMES[i], "rel err", round(rel, 4))
        assert rel < 0.15, (i, ops.TENSOR_NAMES[i], rel)


# (the mode's quality gate: tests/test_psnr_gpu.py -- 320 seeds on the SURVEY 8(d) scene, with and without the feature loss)


@pytest.mark.parametrize("feat", [False, True])
@pytest.mark.parametrize("net", [(1, 600, 128), (2, 96, 256)])
def test_bf16_background_step_close_to_fp32(dev, feat, net):
    """The layer-wise path with bf16 GEMM operands against its fp32 self: hidden 128 (the background network) and
    hidden 256 with >= 4096 samples per object (configs[4]: resident-panel GEMMs, the concatenated layers as one
    contraction, activations stored in bf16 -- sized with the 16-bit bit of objnerf_train_workspace_bytes)."""
    from openobj_amd import init as obj_init
    K, R, H = net
    n1, n2 = 16, 48
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, H, 512, seed=3))
    arena.scale.fill_(5.0)
    b = synthetic.random_batch(K, R, n1, n2, seed=12, feat_dim=512 if feat else 0)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    batch = {k: T(b[k]).to(dev) for k in keys}
    ws32 = ops.TrainWorkspace(arena, K, R, n1 + n2, feat)
    ws16 = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, precision="bf16")
    if H == 256:
        assert ws16.nbytes < (0.8 if feat else 0.75) * ws32.nbytes          # h1 .. hc and d_hc .. d_h1 in 16 bit
    ops.train_step(arena, ws32, batch, with_feat=feat)
    ops.train_step(arena, ws16, batch, with_feat=feat, bf16=True)
    torch.cuda.synchronize()
    assert int(ws16.status.item()) == 0 and bool(torch.isfinite(ws16.grads).all())
    np.testing.assert_allclose(ws16.loss_terms.cpu(), ws32.loss_terms.cpu(), rtol=3e-2, atol=3e-3)
    g32, g16 = arena.views(ws32.grads), arena.views(ws16.grads)
    for i in range(19):
        if i in ops.FEAT_TENSORS and not feat:
            continue
        a, r = g16[i].double().cpu(), g32[i].double().cpu()
        rel = float((a - r).norm() / (r.norm() + 1e-12))
        assert rel < 0.15, (i, ops.TENSOR_NAMES[i], rel)


def test_bf16_second_generation_kernel_edge_cases(golden, dev):
    """The 64-sample bf16 kernel (objnerf_train_bf16v2.hip) on what the first generation was tested for at other shapes:
    (a) the cross-object early return (render_rays.py:89-94): an object without a label-1 ray zeroes the depth / colour
    terms and their gradients of EVERY object -- exact zeros, and the opacity term still flows; (b) the batch given as
    origins / dirs / z (the seeded sampler's compact form) equals the batch given as points bit for bit; (c) global
    flags / counts handed in (object sharding) are honoured; (d) the step is bit-reproducible."""
    K, R, n1, n2 = 3, 130, 16, 48
    arena = _arena(golden, K, dev)
    b = synthetic.random_batch(K, R, n1, n2, seed=77)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"]
    batch = {k: T(b[k]).to(dev) for k in keys}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False)
    ops.train_step(arena, ws, batch, bf16=True)
    torch.cuda.synchronize()
    g0, t0 = ws.grads.clone(), ws.loss_terms.clone()
    ops.train_step(arena, ws, batch, bf16=True)
    torch.cuda.synchronize()
    assert torch.equal(ws.grads, g0) and torch.equal(ws.loss_terms, t0)                       # (d)
    b2 = {k: T(b[k]).to(dev) for k in ["origins", "dirs", "z", "gt_depth", "gt_rgb", "labels"]}
    ops.train_step(arena, ws, b2, bf16=True)
    torch.cuda.synchronize()
    assert torch.equal(ws.grads, g0)                                                          # (b)
    lab = batch["labels"].clone()
    lab[1][lab[1] == 1] = 0
    ops.train_step(arena, ws, dict(batch, labels=lab), bf16=True)
    torch.cuda.synchronize()
    t = ws.loss_terms.cpu()
    assert float(t[:, :2].abs().max()) == 0.0 and float(t[:, 2].min()) > 0.0                  # (a)
    gv = arena.views(ws.grads)
    for i in (10, 11, 12, 13):                                     # the colour branch sees no loss term at all
        assert float(gv[i].abs().max()) == 0.0, ops.TENSOR_NAMES[i]
    assert float(gv[0].abs().max()) > 0.0
    flags = torch.tensor([1, 0], dtype=torch.int32, device=dev)    # (c) the same decision handed in from outside
    counts = ops.label_counts(batch["labels"])[0]
    ops.train_step(arena, ws, batch, bf16=True, global_flags=flags, global_counts=counts)
    torch.cuda.synchronize()
    assert float(ws.loss_terms[:, :2].abs().max()) == 0.0
    assert float(arena.views(ws.grads)[12].abs().max()) == 0.0


def test_bf16_second_generation_feature_kernel_edge_cases(golden, dev):
    """The same four properties for the feature-loss instantiation (train_fused_bf16v2f_kernel: configs[2] / [3] in
    bf16): bit-reproducible, points == origins / dirs / z bit for bit, and the early return (loss.py:81-99 sits behind the
    same render_rays.py:89-94 decision) zeroes the depth, colour AND feature terms with the feature branch's gradients --
    exact zeros -- while the opacity term still trains the trunk.  A ragged ray count (130 = 65 tiles, the last one's
    second ray past the end of nothing: R is even; 131 leaves a half-filled tile) runs too."""
    for R in (130, 131):
        K, n1, n2 = 3, 16, 48
        arena = _arena(golden, K, dev)
        b = synthetic.random_batch(K, R, n1, n2, seed=77 + R, feat_dim=512)
        keys = ["pts", "z", "gt_depth", "gt_rgb", "labels", "gt_feat"]
        batch = {k: T(b[k]).to(dev) for k in keys}
        ws = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
        ops.train_step(arena, ws, batch, with_feat=True, bf16=True)
        torch.cuda.synchronize()
        assert int(ws.status.item()) == 0 and bool(torch.isfinite(ws.grads).all())
        g0, t0 = ws.grads.clone(), ws.loss_terms.clone()
        assert float(t0[:, 3].min()) > 0.0
        ops.train_step(arena, ws, batch, with_feat=True, bf16=True)
        torch.cuda.synchronize()
        assert torch.equal(ws.grads, g0) and torch.equal(ws.loss_terms, t0)
        b2 = {k: T(b[k]).to(dev) for k in ["origins", "dirs", "z", "gt_depth", "gt_rgb", "labels", "gt_feat"]}
        ops.train_step(arena, ws, b2, with_feat=True, bf16=True)
        torch.cuda.synchronize()
        assert torch.equal(ws.grads, g0)
        # against the fp32 fused kernel at this ragged shape
        ws32 = ops.TrainWorkspace(arena, K, R, n1 + n2, True)
        ops.train_step(arena, ws32, batch, with_feat=True)
        torch.cuda.synchronize()
        np.testing.assert_allclose(t0.cpu(), ws32.loss_terms.cpu(), rtol=2e-2, atol=2e-3)
        g32, g16 = arena.views(ws32.grads), arena.views(g0)
        for i in range(19):
            a, r = g16[i].double().cpu(), g32[i].double().cpu()
            assert float((a - r).norm() / (r.norm() + 1e-12)) < 0.15, (R, i, ops.TENSOR_NAMES[i])
        lab = batch["labels"].clone()
        lab[1][lab[1] == 1] = 0
        ops.train_step(arena, ws, dict(batch, labels=lab), with_feat=True, bf16=True)
        torch.cuda.synchronize()
        t = ws.loss_terms.cpu()
        assert float(t[:, :2].abs().max()) == 0.0 and float(t[:, 3].abs().max()) == 0.0 and float(t[:, 2].min()) > 0.0
        gv = arena.views(ws.grads)
        for i in (10, 11, 12, 13) + tuple(ops.FEAT_TENSORS):
            assert float(gv[i].abs().max()) == 0.0, ops.TENSOR_NAMES[i]
        assert float(gv[0].abs().max()) > 0.0
        # the same decision handed in from outside (object sharding: global flags / counts)
        flags = torch.tensor([1, 0], dtype=torch.int32, device=dev)
        counts = ops.label_counts(batch["labels"])[0]
        ops.train_step(arena, ws, batch, with_feat=True, bf16=True, global_flags=flags, global_counts=counts)
        torch.cuda.synchronize()
        assert float(ws.loss_terms[:, :2].abs().max()) == 0.0 and float(ws.loss_terms[:, 3].abs().max()) == 0.0
        assert float(arena.views(ws.grads)[16].abs().max()) == 0.0
        # ... and with nothing empty, the global counts normalise the terms: twice the counts, half the gradient
        flags0 = torch.zeros(2, dtype=torch.int32, device=dev)
        ops.train_step(arena, ws, batch, with_feat=True, bf16=True, global_flags=flags0, global_counts=counts)
        torch.cuda.synchronize()
        assert torch.equal(ws.grads, g0)
        ops.train_step(arena, ws, batch, with_feat=True, bf16=True, global_flags=flags0, global_counts=2 * counts)
        torch.cuda.synchronize()
        ref0 = arena.views(g0)[0].double()
        sel = ref0.abs() > 1e-6
        assert float((arena.views(ws.grads)[0].double()[sel] / ref0[sel] - 0.5).abs().max()) < 1e-3


@pytest.mark.parametrize("feat", [False, True])
def test_bf16_second_generation_kernels_shape_sweep(golden, dev, feat):
    """Ragged and degenerate launch shapes of the 64-sample bf16 kernels against the fp32 fused kernel: a single ray (half
    a tile), odd ray counts around one and two tiles, one object, more objects than CUs with a few rays each (workgroups
    that own ONE tile, or none).  Finite, status clean, losses and every gradient tensor within the mode's noise."""
    n1, n2 = 16, 48
    for K, R in ((1, 1), (1, 2), (2, 3), (1, 63), (3, 65), (2, 127), (1, 129), (257, 2), (300, 5)):
        arena = _arena(golden, K, dev)
        b = synthetic.random_batch(K, R, n1, n2, seed=1000 + 7 * K + R, feat_dim=512 if feat else 0)
        keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
        batch = {k: T(b[k]).to(dev) for k in keys}
        ws32 = ops.TrainWorkspace(arena, K, R, n1 + n2, feat)
        ws16 = ops.TrainWorkspace(arena, K, R, n1 + n2, feat)
        ops.train_step(arena, ws32, batch, with_feat=feat)
        ops.train_step(arena, ws16, batch, with_feat=feat, bf16=True)
        torch.cuda.synchronize()
        assert int(ws16.status.item()) == 0 and bool(torch.isfinite(ws16.grads).all()), (K, R)
        ncol = 4 if feat else 3
        np.testing.assert_allclose(ws16.loss_terms.cpu()[:, 1:ncol], ws32.loss_terms.cpu()[:, 1:ncol], rtol=3e-2, atol=3e-3)
        g32, g16 = arena.views(ws32.grads), arena.views(ws16.grads)
        for i in (range(19) if feat else list(range(14)) + [18]):
            a, r = g16[i].double().cpu(), g32[i].double().cpu()
            # few rays: one L1 sign or ReLU branch decided the other way moves a whole row -- bound the norm, loosely
            bound = 0.5 if K * R < 8 else (0.25 if K * R < 64 else 0.15)
            # (+ an absolute floor: with three rays a saturated batch leaves gradients of 1e-6 and below)
            assert float((a - r).norm()) < bound * float(r.norm()) + 1e-4, (K, R, ops.TENSOR_NAMES[i])
