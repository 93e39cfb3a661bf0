"""GPU: the opt-in 16-bit operand modes against THEIR specification, not against the build's own fp32 step.

OBJNERF_TRAIN_BF16 / OBJNERF_TRAIN_FP16 are not the reference's arithmetic (fp32, train.py:74), so the 1e-4 parity bar
does not apply to them; what they claim to compute is model.py:61-103 with both operands of every hidden nn.Linear
rounded to the operand type and fp32 accumulation.  oracle.mlp_forward_stacked_16 states exactly that (rounding points
of the layer-wise path: activations stored in the operand type at hidden 256 with >= 4096 samples, rounded at the GEMM
otherwise; back-propagated gradients rounded as the next GEMM's operand, fp16 pre-scaled by 2^(floor(log2 R) + 3)).
A kernel with a wrong term in a small tensor cannot hide inside these bounds the way it could inside the former
"within 2 - 15 % of the fp32 kernel" ones: what remains is the order of fp32 accumulation and a handful of ReLU ties.
"""
import math

import numpy as np
import pytest
import torch

from conftest import T
from openobj_amd import init as obj_init
from openobj_amd import ops, synthetic
from parity_util import oracle_step_16, rel_norm

pytestmark = pytest.mark.gpu

DT = {"bf16": torch.bfloat16, "fp16": torch.float16}
# relative-norm bounds per gradient tensor: an order below the former "within 15 % / 2 % of the fp32 kernel" ones
# (measured, profiles/r03_16bit_spec.txt: bf16 <= 3.3e-3, fp16 <= 7.1e-3 -- the colour layer of a 9000-sample batch,
# where a handful of units whose pre-activation sits within one 16-bit rounding step of zero take the other ReLU branch
# in any two implementations of the same specification; most tensors 1e-5 .. 1e-3)
BOUND = {"bf16": 0.01, "fp16": 0.01}


def _run(dev, K, R, n1, n2, H, feat, mode, seed=7):
    arena = ops.ParamArena(K, ops.NetShape(H, 512, 6), dev)
    st = obj_init.init_stacked(K, H, 512, seed=seed)
    arena.load_stacked(st)
    b = synthetic.random_batch(K, R, n1, n2, seed=5 + R, feat_dim=512 if feat else 0)
    keys = ["pts", "z", "gt_depth", "gt_rgb", "labels"] + (["gt_feat"] if feat else [])
    batch = {k: T(b[k]).to(dev) for k in keys}
    layerwise = (H != 32) or (n1 + n2 > 64) or mode == "fp16"
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, feat, precision=mode)
    ops.train_step(arena, ws, batch, with_feat=feat, bf16=mode)
    torch.cuda.synchronize()
    assert int(ws.status.item()) == 0 and bool(torch.isfinite(ws.grads).all())
    return arena, st, b, ws, layerwise


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 64, 16, 48, 256), (1, 4096, 16, 48, 256), (1, 64, 32, 96, 256), (2, 200, 5, 9, 128),
                                   (1, 4096, 16, 48, 64),
                                   # fused hidden-256 path: S = 32 with a ragged ray count, workgroups that cross from one
                                   # object's tiles (and weight image) to the next one's
                                   (3, 700, 8, 24, 256), (2, 300, 16, 48, 256), (2, 101, 8, 24, 256), (1, 33, 16, 48, 256)])
def test_layerwise_16bit_step_matches_its_specification(dev, mode, shape):
    """Layer-wise path (any width), no feature loss -- configs[4]'s arithmetic: hidden 256 with R x S >= 4096 samples
    per object runs the resident-panel GEMMs with activations stored in the operand type (act16)."""
    K, R, n1, n2, H = shape
    if mode == "fp16" and H == 128 and R * (n1 + n2) < 20000:
        pytest.skip("small hidden-128 batches run the one-launch fp32 kernels in fp16 mode (objnerf_generic.hip "
                    "small_batch_rt): nothing is rounded, test_hip_parity.py covers that path")
    arena, st, b, ws, _ = _run(dev, K, R, n1, n2, H, False, mode)
    # hidden 256 with S a power of two in 32..256: the fused objnerf_train256.hip path -- activations ARE the next MFMA's
    # 16-bit operands, the heads and their gradients run on the matrix core too
    fused256 = H == 256 and (n1 + n2) in (32, 64, 128)
    gs = 2.0 ** (math.floor(math.log2(R)) + 3) if mode == "fp16" else 1.0
    o = oracle_step_16(list(st[:18]), st[18], 2.0, b, False, DT[mode], fused256, gs, device=dev,
                       round_head_weights=fused256, round_head_grads=fused256)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, :3], o["terms"][:, :3], rtol=2e-4 if mode == "fp16" else 2e-3,
                               atol=1e-5)
    gv = arena.views(ws.grads)
    worst = 0.0
    for i in list(range(14)) + [18]:
        rel = rel_norm(gv[i], o["grads"][i])
        worst = max(worst, rel)
        print(f"{mode} H={H} R={R} {ops.TENSOR_NAMES[i]:24s} rel {rel:.2e}")
        assert rel < BOUND[mode], (i, ops.TENSOR_NAMES[i], rel)
    for i in ops.FEAT_TENSORS:
        assert float(gv[i].abs().max()) == 0.0
    print(f"worst {worst:.2e}")


@pytest.mark.parametrize("mode", ["bf16", "fp16"])
@pytest.mark.parametrize("shape", [(2, 80, 16, 48, 256), (2, 200, 5, 9, 128)])
def test_layerwise_16bit_feature_step_matches_its_specification(dev, mode, shape):
    """With the 512-d feature loss.  The kernels apply the C x H head in the hoisted form: its Gram
    matrix and the per-ray u = W_of^T g are GEMMs too, so W_of AND the target features enter rounded; the
    specification rounds W_of only (round_head_weights), hence the wider bound on the feature branch's tensors."""
    K, R, n1, n2, H = shape
    if mode == "fp16" and H == 128:
        pytest.skip("small hidden-128 batches: fp32 kernels in fp16 mode")
    arena, st, b, ws, _ = _run(dev, K, R, n1, n2, H, True, mode)
    act16 = H == 256 and R * (n1 + n2) >= 4096
    gs = 2.0 ** (math.floor(math.log2(R)) + 3) if mode == "fp16" else 1.0
    o = oracle_step_16(list(st[:18]), st[18], 2.0, b, True, DT[mode], act16, gs, device=dev)
    gv = arena.views(ws.grads)
    for i in range(19):
        rel = rel_norm(gv[i], o["grads"][i])
        print(f"{mode} H={H} feat {ops.TENSOR_NAMES[i]:24s} rel {rel:.2e}")
        assert rel < (4 if i in ops.FEAT_TENSORS else 2) * BOUND[mode], (i, ops.TENSOR_NAMES[i], rel)


# Second-generation kernels (64 samples per ray), per tensor: what tools/bf16v2_diag.py / bf16v2f_diag.py print, ASSERTED
# (round 5).  Measured relative Frobenius distance to the operand-rounded specification at 3 x 300 x 64 = 57 600 samples
# and 2 x 333 x 64 (the larger of the two; profiles/r05_bf16_spec_rel.txt), in units of 1e-3, without / with the feature
# loss; the bound of a tensor is 1.6 x its measured distance (at least 0.5 %, at most V2_CAP).  Round 4's hipcc miscompile of the packed ReLU (gradients off
# by 10 - 170 %) passes none of them; neither does a 4 % error in one tensor that the former flat 6 - 8 % bounds let by.
V2_MEASURED = {"in_layer.0.weight": (17.7, 17.5), "in_layer.0.bias": (8.1, 7.1), "mid1.0.0.weight": (9.1, 8.7),
               "mid1.0.0.bias": (6.9, 6.6), "cat_layer.0.weight": (7.2, 7.2), "cat_layer.0.bias": (2.8, 2.6),
               "mid2.0.0.weight": (3.0, 2.8), "mid2.0.0.bias": (1.8, 1.6), "out_alpha.weight": (1.3, 1.3),
               "out_alpha.bias": (0.7, 0.7), "color_linear.0.weight": (15.9, 15.7), "color_linear.0.bias": (9.4, 9.2),
               "out_color.weight": (1.3, 1.1), "out_color.bias": (0.5, 0.2), "clip_linear.0.weight": (None, 25.4),
               "clip_linear.0.bias": (None, 22.4), "out_clip.weight": (None, 2.2), "out_clip.bias": (None, 0.6),
               "B_layer.weight": (22.0, 21.9)}
V2_CAP = {False: 0.03, True: 0.04}      # trunk tensors <= 3 %, feature-branch tensors (and d B) <= 4 %: the review's figures


def v2_bound(name, i, feat, samples_per_object):
    """Bound of tensor `name` for a second-generation launch: the table above at >= 6400 samples per object (the sums
    behind a gradient are then long enough for the rounding noise to average as in the measured shape); below that --
    (60, 40, ..): 2560 per object, (300, 9, ..): 576 -- a flat 6 % (measured <= 3.4 %; 10 - 12 % until round 5)."""
    if samples_per_object < 6400:
        return 0.06
    cap = V2_CAP[i in ops.FEAT_TENSORS or name == "B_layer.weight"]
    m = V2_MEASURED[name][1 if feat else 0]
    return min(cap, max(0.005, 1.6e-3 * m))


@pytest.mark.parametrize("shape", [(3, 300, 16, 48), (2, 211, 5, 9), (4, 128, 8, 24), (2, 333, 16, 48), (300, 9, 16, 48)])
def test_fused_bf16_kernel_matches_its_specification(dev, shape):
    """The fused hidden-32 kernel in bf16 mode (objnerf_train_bf16.hip, the kernel behind BASELINE configs[1] / [2]'s
    dtype): activations are packed to bf16 as the next MFMA's operand (act16 semantics: every consumer sees the rounded
    value), the heads run in fp32.  Measured (tools/bf16_fused_diag.py, profiles/r03_bf16_fused_diag.txt): this kernel sits
    1 - 4 % from the specification per tensor (the layer-wise bf16 path: 0.1 - 0.3 %), against 4 - 6 % between the
    specification and fp32 -- its transcendental-unit sin / cos / exp and its own rounding points are not modelled.
    Bound 6 % (10 % below 20 000 samples; the former bound against the fp32 kernel was 15 %).
    Round 4: the 64-sample shapes run the SECOND-generation kernel (objnerf_train_bf16v2.hip: every tensor within 2 % on
    the first shape, tools/bf16v2_diag.py; its head gradients are rounded as weight-gradient operands: round_head_grads);
    (2, 333, ..) ends on a half-filled tile, (300, 9, ..) has more objects than CUs and 4.5 tiles per object."""
    K, R, n1, n2 = shape
    arena, st, b, ws, _ = _run(dev, K, R, n1, n2, 32, False, "bf16", seed=11)
    o = oracle_step_16(list(st[:18]), st[18], 2.0, b, False, torch.bfloat16, True, 1.0, device=dev,
                       round_head_grads=(n1 + n2 == 64))
    # (the depth term divides by sqrt(var) + 1e-4: a ray whose weight sits on one sample amplifies a 1e-6 difference)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, 1:3], o["terms"][:, 1:3], rtol=5e-3, atol=1e-4)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, 0], o["terms"][:, 0], rtol=5e-2, atol=1e-3)
    gv = arena.views(ws.grads)
    for i in list(range(14)) + [18]:
        rel = rel_norm(gv[i], o["grads"][i])
        print(f"fused bf16 R={R} {ops.TENSOR_NAMES[i]:24s} rel {rel:.2e}")
        if n1 + n2 == 64:           # second generation: per-tensor bounds
            bound = v2_bound(ops.TENSOR_NAMES[i], i, False, R * 64)
        else:                       # first generation (S != 64): measured 3.0 % (S = 14) / 5.5 % (S = 32, 16 384 samples)
            bound = 0.06 if K * R * (n1 + n2) >= 20000 else 0.10
        assert rel < bound, (i, ops.TENSOR_NAMES[i], rel, bound)


@pytest.mark.parametrize("shape", [(3, 300, 16, 48), (2, 333, 16, 48), (60, 40, 16, 48), (300, 9, 16, 48)])
def test_fused_bf16_feature_kernel_matches_its_specification(dev, shape):
    """The feature-loss instantiation of the second-generation kernel (train_fused_bf16v2f_kernel, configs[2] / [3] in
    bf16) against the operand-rounded specification with the 512-d term.  Measured (tools/bf16v2f_diag.py, 57 600
    samples): trunk tensors within 1.2 %, the feature layer's 2.5 % (the hoisted head's Gram matrix and u = W_of^T g are
    bf16-operand GEMMs here, the specification rounds W_of only), the head weights' gradients -- per-lane fp32 sums of
    unrounded head gradients in this instantiation, so NOT round_head_grads -- within 0.25 %.  (2, 333, ..) ends on a
    half-filled tile; (60, 40, ..) has 20 tiles per object and fewer tiles than workgroups would like."""
    K, R, n1, n2 = shape
    arena, st, b, ws, _ = _run(dev, K, R, n1, n2, 32, True, "bf16", seed=11)
    o = oracle_step_16(list(st[:18]), st[18], 2.0, b, True, torch.bfloat16, True, 1.0, device=dev)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, 1:], o["terms"][:, 1:], rtol=5e-3, atol=1e-4)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, 0], o["terms"][:, 0], rtol=5e-2, atol=1e-3)
    gv = arena.views(ws.grads)
    for i in range(19):
        rel = rel_norm(gv[i], o["grads"][i])
        print(f"fused bf16 feat R={R} {ops.TENSOR_NAMES[i]:24s} rel {rel:.2e}")
        bound = v2_bound(ops.TENSOR_NAMES[i], i, True, R * 64)
        assert rel < bound, (i, ops.TENSOR_NAMES[i], rel, bound)


@pytest.mark.parametrize("feat", [False, True])
def test_fused_bf16_kernels_at_the_full_baseline_size(dev, feat):
    """BASELINE configs[1] / [2] at their full size -- 50 objects x 4096 rays x 64 samples, 13.1 M samples -- in bf16
    mode against the operand-rounded specification evaluated on the GPU (torch, fp32 accumulation: the same anchor
    test_fp16_gpu.py uses for configs[4]).  The bounds are the small-shape ones; at this size every tensor sits well
    inside them (the relative rounding noise of a sum falls with its length)."""
    K, R, n1, n2 = 50, 4096, 16, 48
    arena, st, b, ws, _ = _run(dev, K, R, n1, n2, 32, feat, "bf16", seed=23)
    o = oracle_step_16(list(st[:18]), st[18], 2.0, b, feat, torch.bfloat16, True, 1.0, device=dev,
                       round_head_grads=not feat)
    cols = slice(1, 4 if feat else 3)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, cols], o["terms"][:, cols], rtol=5e-3, atol=1e-4)
    np.testing.assert_allclose(ws.loss_terms.double().cpu()[:, 0], o["terms"][:, 0], rtol=5e-2, atol=1e-3)
    gv = arena.views(ws.grads)
    worst = 0.0
    for i in (range(19) if feat else list(range(14)) + [18]):
        rel = rel_norm(gv[i], o["grads"][i])
        worst = max(worst, rel)
        print(f"full size feat={feat} {ops.TENSOR_NAMES[i]:24s} rel {rel:.2e}")
        # (round 5: the review's <= 3 % trunk / <= 4 % feature-branch and d B; measured worst 1.2 % / 1.7 % / 2.9 %)
        name = ops.TENSOR_NAMES[i]
        assert rel < V2_CAP[i in ops.FEAT_TENSORS or name == "B_layer.weight"], (i, name, rel)
    print(f"worst {worst:.2e}")


def test_fused_kernel_embedding_rows(dev, golden):
    """The fused fp32 kernel never materialises its embedding; objnerf_train_args.emb_debug makes its tiles write the rows
    they formed in registers.  Against fixture G1 (the reference's UniDirsEmbed): 2e-5 -- G1's arguments reach 1e2,
    where one ulp of the fp32 argument (which the kernel reproduces: fl(fl(p 2^f) pi) = 2^f fl(p pi)) moves sin by
    1e-5.  Against the standalone embedding kernel (accurate sin of the SAME fp32 argument, itself pinned to G1 by
    test_embed_g1): 5e-6 (measured 3.7e-6: octave 5 is two angle doublings past its anchor, objnerf_mlp32.h)."""
    g = golden("g1_embed")
    for tag, scale in (("s2", 2.0), ("s5", 5.0)):
        pts = T(g[f"pts_{tag}"]).reshape(1, 10, 7, 3).to(dev)
        arena = ops.ParamArena(1, ops.NetShape(), dev)
        arena.load_stacked(obj_init.init_stacked(1, 32, 512, seed=1))
        arena.views()[18].copy_(T(g[f"B_{tag}"]).to(dev)[None])
        arena.scale.fill_(scale)
        batch = {"pts": pts, "z": torch.rand(1, 10, 7, device=dev).sort(-1).values + 0.5,
                 "gt_depth": torch.ones(1, 10, device=dev), "gt_rgb": torch.rand(1, 10, 3, device=dev),
                 "labels": torch.ones(1, 10, dtype=torch.uint8, device=dev)}
        ws = ops.TrainWorkspace(arena, 1, 10, 7, False)
        emb = torch.full((1, 10, 7, 129), float("nan"), device=dev)
        ops.train_step(arena, ws, batch, emb_debug=emb)
        ref = ops.embed(arena, pts.reshape(1, -1, 3)).reshape(1, 10, 7, 129)
        torch.cuda.synchronize()
        assert bool(torch.isfinite(emb).all())                                   # every entry of every row was written
        assert float((emb.cpu().reshape(2, 5, 7, 129) - T(g[f"emb_{tag}"])).abs().max()) < 2.3e-5
        assert float((emb - ref).abs().max()) < 5e-6
        plain = ws.grads.clone()
        ops.train_step(arena, ws, batch)                                         # the hook changes nothing else
        torch.cuda.synchronize()
        assert torch.equal(ws.grads, plain)
    # training-shaped points (|p| <= 4 m), ragged tile, several objects
    K, R, n1, n2 = 3, 37, 4, 12
    arena = ops.ParamArena(K, ops.NetShape(), dev)
    arena.load_stacked(obj_init.init_stacked(K, 32, 512, seed=2))
    b = synthetic.random_batch(K, R, n1, n2, seed=77)
    batch = {k: T(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
    ws = ops.TrainWorkspace(arena, K, R, n1 + n2, False)
    emb = torch.full((K, R, n1 + n2, 129), float("nan"), device=dev)
    ops.train_step(arena, ws, batch, emb_debug=emb)
    ref = ops.embed(arena, batch["pts"].reshape(K, -1, 3)).reshape(K, R, n1 + n2, 129)
    torch.cuda.synchronize()
    err = float((emb - ref).abs().max())
    print("fused embedding vs standalone kernel, max abs", err)
    assert bool(torch.isfinite(emb).all()) and err < 5e-6


def test_emb_debug_is_refused_off_the_fused_fp32_path(dev):
    arena = ops.ParamArena(1, ops.NetShape(64, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(1, 64, 512, seed=1))
    b = synthetic.random_batch(1, 16, 2, 6, seed=1)
    batch = {k: T(b[k]).to(dev) for k in ["pts", "z", "gt_depth", "gt_rgb", "labels"]}
    ws = ops.TrainWorkspace(arena, 1, 16, 8, False)
    with pytest.raises(ops.ObjnerfError):
        ops.train_step(arena, ws, batch, emb_debug=torch.zeros(1, 16, 8, 129, device=dev))
