"""GPU: the reference's loop body kept verbatim (train.py:424-436) -- vmap(pe_model) -> vmap(fc_model) ->
loss.step_batch_loss -> loss.backward() -- on the mirrored modules: autograd reaches the stacked parameters through
objnerf_mlp_backward_ws / objnerf_embed_bwd and yields the reference's gradients (fixture G5)."""
import numpy as np
import pytest
import torch

from conftest import T
from oracle import objnerf_oracle as O
from openobj_amd import loss as oloss
from openobj_amd import ops, synthetic, trainer, utils
from test_api_gpu import make_cfg, make_trainers, oracle_params
from test_hip_parity import arena_from_fixture, maxerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag", ["s10_nofeat", "s10_feat", "s64_nofeat"])
def test_reference_loop_body_g5(golden, dev, tag):
    g = golden(f"g5_step_{tag}")
    K, R, n1, n2, feat_on = [int(x) for x in g["meta"]]
    arena = arena_from_fixture(g, dev)
    views = arena.views()
    fc_model, fc_param, fc_buffer = utils.StackedModel(arena, "fc"), [v.requires_grad_() for v in views[:18]], []
    pe_model, pe_param, pe_buffer = utils.StackedModel(arena, "pe"), [views[18].requires_grad_()], []
    b = {k: T(v).to(dev) for k, v in synthetic.random_batch(K, R, n1, n2, seed=500, feat_dim=512).items()}
    # ---- train.py:424-436
    batch_embedding = utils.vmap(pe_model)(pe_param, pe_buffer, b["pts"])
    batch_alpha, batch_color, batch_clip = utils.vmap(fc_model)(fc_param, fc_buffer, batch_embedding)
    kw = dict(gt_partfeat=b["gt_feat"], pred_partfeat=batch_clip) if feat_on else {}
    batch_loss, _ = oloss.step_batch_loss(batch_alpha, batch_color, b["gt_depth"], b["gt_rgb"], b["labels"],
                                          torch.ones_like(b["gt_depth"], dtype=torch.bool), b["z"], **kw)
    batch_loss.backward()
    # ----
    assert abs(batch_loss.item() - g["loss"][0]) < 1e-4 * abs(g["loss"][0])
    for i, p in enumerate(fc_param + pe_param):
        if i in ops.FEAT_TENSORS and not feat_on:
            assert p.grad is None
            continue
        ref = g[f"grad0_{i}"]
        scale = max(1e-3, float(np.abs(ref).max()))
        assert maxerr(p.grad, ref) < 1e-4 * scale, (i, ops.TENSOR_NAMES[i], maxerr(p.grad, ref), scale)


@pytest.mark.parametrize("hidden", [32, 128])
def test_single_module_autograd_vs_oracle(dev, hidden):
    """pe(x) -> fc_occ_map(emb) -> a scalar -> backward() on ONE Trainer's modules (the background network of
    train.py:449-452 at hidden 128), against torch autograd over the oracle."""
    torch.manual_seed(11)
    c = make_cfg(dev)
    if hidden != 32:                                   # train.py:213-216: the background Trainer is built on a copy of the
        c.obj_id = 0                                   # config with hidden_feature_size = hidden_feature_size_bg
        c.hidden_feature_size = c.hidden_feature_size_bg
    t = trainer.Trainer(c)
    assert t.fc_occ_map._arena.net.hidden == hidden
    rs = np.random.RandomState(3)
    pts = torch.from_numpy(rs.uniform(-2, 2, (40, 9, 3)).astype(np.float32))
    wa, wc, wf = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in ((40, 9, 1), (40, 9, 3), (40, 9, 512))]
    emb = t.pe(pts.to(dev))
    alpha, color, clip = t.fc_occ_map(emb)
    ((alpha * wa.to(dev)).sum() + (color * wc.to(dev)).sum() + 0.01 * (clip * wf.to(dev)).sum()).backward()
    fc, B = oracle_params([t])
    fc = [p[0].clone().requires_grad_() for p in fc]
    B0 = B[0].clone().requires_grad_()
    a, cc, f = O.mlp_forward(fc, O.unidirs_embed(pts, B0, float(t.pe.scale)))
    ((a * wa).sum() + (cc * wc).sum() + 0.01 * (f * wf).sum()).backward()
    mine = [p.grad for p in t.fc_occ_map.parameters()] + [t.pe.B_layer.weight.grad]
    for i, (m, r) in enumerate(zip(mine, fc + [B0])):
        scale = max(1e-3, float(r.grad.abs().max()))
        assert maxerr(m, r.grad) < 1e-4 * scale, (i, maxerr(m, r.grad), scale)


@pytest.mark.parametrize("hidden", [128, 256])
def test_stacked_autograd_wide_networks_vs_oracle(dev, hidden):
    """vmap(pe) -> vmap(fc) -> weighted sums -> backward() for K = 2 stacked networks of the background / stress widths
    (the layer-wise backward entry), against torch autograd over the stacked oracle."""
    from openobj_amd import init as obj_init
    K, N = 2, 300
    st = obj_init.init_stacked(K, hidden, 512, seed=21)
    arena = ops.ParamArena(K, ops.NetShape(hidden, 512, 6), dev)
    arena.load_stacked(st)
    arena.scale.fill_(2.0)
    views = arena.views()
    fc_param, pe_param = [v.requires_grad_() for v in views[:18]], [views[18].requires_grad_()]
    rs = np.random.RandomState(8)
    pts = torch.from_numpy(rs.uniform(-2, 2, (K, N, 3)).astype(np.float32))
    wa, wc, wf = [torch.from_numpy(rs.standard_normal(s).astype(np.float32)) for s in ((K, N, 1), (K, N, 3), (K, N, 512))]
    emb = utils.vmap(utils.StackedModel(arena, "pe"))(pe_param, [], pts.to(dev))
    alpha, color, clip = utils.vmap(utils.StackedModel(arena, "fc"))(fc_param, [], emb)
    ((alpha * wa.to(dev)).sum() + (color * wc.to(dev)).sum() + 0.01 * (clip * wf.to(dev)).sum()).backward()
    ref = [t.clone().requires_grad_() for t in st]
    e = O.embed_stacked(ref[18], torch.full((K,), 2.0), pts)
    a, c, f = O.mlp_forward_stacked(ref[:18], e, True)
    ((a * wa).sum() + (c * wc).sum() + 0.01 * (f * wf).sum()).backward()
    for i, (m, r) in enumerate(zip(fc_param + pe_param, ref)):
        scale = max(1e-3, float(r.grad.abs().max()))
        assert maxerr(m.grad, r.grad) < 1e-4 * scale, (i, ops.TENSOR_NAMES[i], maxerr(m.grad, r.grad), scale)


def test_inference_calls_build_no_graph(dev):
    t = make_trainers(1, dev, 5)[0]
    with torch.no_grad():
        alpha, color, clip = t.fc_occ_map(t.pe(torch.zeros(4, 3, device=dev)))
    assert alpha.grad_fn is None and not alpha.requires_grad


def test_backward_after_the_arena_changed_raises(dev):
    """backward() recomputes from the live arena: an optimiser step, a copy into a parameter or a scale change between
    forward and backward must raise (torch does on a modified saved tensor), a repeated forward must not."""
    from openobj_amd import optim as ooptim
    torch.manual_seed(5)
    t = trainer.Trainer(make_cfg(dev))
    pts = torch.from_numpy(np.random.RandomState(1).uniform(-1, 1, (6, 5, 3)).astype(np.float32)).to(dev)

    def fwd():
        a, c, _ = t.fc_occ_map(t.pe(pts))
        return a.sum() + c.sum()

    l1 = fwd()
    fwd()                                                  # a second forward (same scale) does not invalidate the first
    l1.backward()
    assert t.pe.B_layer.weight.grad is not None
    l2 = fwd()
    opt = ooptim.ArenaAdamW(t.arena)
    opt.step(torch.zeros_like(t.arena.params))             # the kernel writes the arena behind torch's back
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        l2.backward()
    # the ops-level writers themselves advance the arena's version (not only optim.ArenaAdamW): a direct
    # ops.adamw_step / adamw_step_flags / train_step(optim=) between forward and backward is caught too
    l2b = fwd()
    ops_mod = __import__("openobj_amd.ops", fromlist=["ops"])
    z = torch.zeros_like(t.arena.params)
    ops_mod.adamw_step(t.arena, z, z.clone(), z.clone(), None, 1, 1e-3, 0.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        l2b.backward()
    l2c = fwd()
    ops_mod.adamw_step_flags(t.arena, z, z.clone(), z.clone(), None, torch.zeros(2, dtype=torch.int32, device=dev),
                             torch.zeros(2, 3, dtype=torch.int32, device=dev), 0, 1e-3, 0.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        l2c.backward()
    l3 = fwd()
    with torch.no_grad():
        list(t.fc_occ_map.parameters())[0].mul_(1.0)       # torch-side in-place write into an arena view
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        l3.backward()
    l4 = fwd()
    t.arena.scale.fill_(3.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        l4.backward()
    other = trainer.Trainer(make_cfg(dev))
    from openobj_amd.autograd import MlpFunction
    with pytest.raises(ValueError, match="not a view of the arena"):
        MlpFunction.apply(t.arena, False, False, t.pe(pts).reshape(1, -1, 129).contiguous(),
                          *list(other.fc_occ_map.parameters()))
