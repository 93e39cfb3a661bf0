"""Pins the CPU oracle (oracle/objnerf_oracle.py) to fixtures produced by running the reference
itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import T
from oracle import objnerf_oracle as O
from openobj_amd import synthetic

TOL = 2e-6


def close(a, b, tol=TOL, rel=0.0):
    a = torch.as_tensor(a).double()
    b = torch.as_tensor(b).double()
    assert a.shape == b.shape, (a.shape, b.shape)
    err = (a - b).abs()
    bound = tol + rel * b.abs()
    assert bool((err <= bound).all()), f"max err {err.max().item():.3e}"


def test_g1_embed(golden):
    g = golden("g1_embed")
    for tag, scale in (("s2", 2.0), ("s5", 5.0)):
        emb = O.unidirs_embed(T(g[f"pts_{tag}"]), T(g[f"B_{tag}"]), scale)
        assert emb.shape[-1] == 129
        close(emb, g[f"emb_{tag}"], 1e-6)


@pytest.mark.parametrize("H", [32, 128])
def test_g2_mlp(golden, H):
    g = golden("g2_mlp")
    p = [T(g[f"h{H}_p{i}"]) for i in range(18)]
    for (name, shape), t in zip(O.param_specs(H), p):
        assert tuple(t.shape) == shape, name
    a, c, f = O.mlp_forward(p, T(g[f"h{H}_emb"]))
    close(a, g[f"h{H}_alpha"], 2e-5)
    close(c, g[f"h{H}_color"], 2e-6)
    close(f, g[f"h{H}_clip"], 2e-5)


def test_g3_render(golden):
    g = golden("g3_render")
    occ = O.occupancy_activation(T(g["alpha"]))
    close(occ, g["occ"], 1e-7)
    tb = O.occupancy_to_termination(occ, is_batch=True)
    close(tb, g["term_b"], 1e-7)
    close(O.occupancy_to_termination(occ[0]), g["term_nb"], 1e-7)
    z = T(g["z"])
    depth = O.render(tb, z)
    close(depth, g["depth"], 1e-6)
    close(O.render(tb, (z - depth[..., None]) ** 2), g["var"], 1e-6)
    close(O.render(tb[..., None], T(g["color"]), dim=-2), g["rgb"], 1e-6)
    close(O.render(tb[..., None], T(g["clip"]), dim=-2), g["feat"], 1e-5)
    close(tb.sum(-1), g["opacity"], 1e-6)


@pytest.mark.parametrize("case", ["normal", "no_label1", "all_unknown"])
@pytest.mark.parametrize("feat_on", [False, True])
def test_g4_loss(golden, case, feat_on):
    g = golden("g4_loss")
    a = T(g["alpha"]).clone().requires_grad_(True)
    c = T(g["color"]).clone().requires_grad_(True)
    f = T(g["clip"]).clone().requires_grad_(True)
    labels = T(g[f"labels_{case}"])
    dmask = torch.ones(labels.shape, dtype=torch.bool)
    kw = dict(gt_partfeat=T(g["gt_feat"]), pred_partfeat=f) if feat_on else {}
    l, _ = O.step_batch_loss(a, c, T(g["gt_depth"]), T(g["gt_rgb"]), labels, dmask, T(g["z"]), **kw)
    tag = f"{case}_{'feat' if feat_on else 'nofeat'}"
    close(l, g[f"loss_{tag}"], 1e-5)
    if l.requires_grad:
        l.backward()
    z0 = lambda t: t.grad if t.grad is not None else torch.zeros_like(t)
    close(z0(a), g[f"dalpha_{tag}"], 1e-6)
    close(z0(c), g[f"dcolor_{tag}"], 1e-6)
    if feat_on:
        close(z0(f), g[f"dclip_{tag}"], 1e-6)
    if case == "all_unknown":
        assert float(l) == 0.0      # every term zeroed for every object (render_rays.py:89-94)


def test_g4_early_return_is_cross_object(golden):
    g = golden("g4_loss")
    assert float(g["loss_no_label1_nofeat"]) < float(g["loss_normal_nofeat"])


def test_g4_var0(golden):
    g = golden("g4_loss")
    labels = T(g["labels_normal"])
    c = T(g["color"]).clone().requires_grad_(True)
    l, _ = O.step_batch_loss(T(g["alpha_var0"]), c, T(g["gt_depth"]), T(g["gt_rgb"]), labels,
                             torch.ones(labels.shape, dtype=torch.bool), T(g["z"]))
    close(l, g["loss_var0"], 0, rel=1e-5)
    l.backward()
    close(c.grad, g["dcolor_var0"], 1e-6)


def _run_oracle_steps(g, n_steps, feat_on, batches):
    fc = [T(g[f"fc0_{i}"]).clone().requires_grad_(True) for i in range(18)]
    B = T(g["B0"]).clone().requires_grad_(True)
    scale = T(g["scale"]) if "scale" in g else torch.full((B.shape[0],), 2.0)
    params = fc + [B]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    rec = dict(loss=[], grads=[], params=[])
    for it in range(n_steps):
        b = batches(it)
        l, _ = O.train_forward_loss(fc, B, scale, T(b["pts"]), T(b["gt_depth"]), T(b["gt_rgb"]),
                                    T(b["labels"]), T(b["z"]),
                                    gt_feat=T(b["gt_feat"]) if feat_on else None)
        grads = torch.autograd.grad(l, params, allow_unused=True)
        rec["loss"].append(l.item())
        rec["grads"].append(grads)
        with torch.no_grad():
            for p, gr, mm, vv in zip(params, grads, m, v):
                if gr is None:
                    continue            # AdamW skips params without grad (no decay either)
                O.adamw_step(p, gr, mm, vv, it + 1, 1e-3, 0.013)
        rec["params"].append([p.detach().clone() for p in params])
    rec["m"], rec["v"] = m, v
    return rec


@pytest.mark.parametrize("tag", ["s10_nofeat", "s10_feat", "s64_feat"])
def test_g5_g6_step_and_adamw(golden, tag):
    g = golden(f"g5_step_{tag}")
    K, R, n1, n2, feat_on = [int(x) for x in g["meta"]]
    rec = _run_oracle_steps(g, 3, bool(feat_on),
                            lambda it: synthetic.random_batch(K, R, n1, n2, seed=500 + it, feat_dim=512))
    np.testing.assert_allclose(rec["loss"], g["loss"], rtol=2e-5)
    for i in range(19):
        gr = rec["grads"][0][i]
        if g["none_grad"][0][i]:
            assert gr is None and i in O.FEAT_PARAM_IDX
        else:
            scale = max(1e-3, float(np.abs(g[f"grad0_{i}"]).max()))
            close(gr, g[f"grad0_{i}"], 2e-5 * scale)
    for it in range(3):
        for i in range(19):
            close(rec["params"][it][i], g[f"param{it}_{i}"], 2e-6)
    for i in range(19):
        close(rec["m"][i], g[f"m_{i}"], 1e-5, rel=1e-4)
        close(rec["v"][i], g[f"v_{i}"], 1e-7, rel=1e-4)


@pytest.mark.parametrize("tag", ["obj", "bg", "metric"])
def test_g7_sample_3d_points(golden, tag):
    g = golden("g7_sample")
    N, M = [int(x) for x in g[f"{tag}_NM"]]
    rgbs = T(g[f"{tag}_rgbs"])
    z, pts, valid, labels = O.sample_3d_points(rgbs[..., 3], T(g[f"{tag}_depth"]), T(g[f"{tag}_origins"]),
                                               T(g[f"{tag}_dirs"]), T(g[f"{tag}_u"]), T(g[f"{tag}_g"]),
                                               N, M, 0.1, 0.05)
    close(z, g[f"{tag}_z"], 1e-6)       # CPU torch.linspace is SIMD-width dependent in the last bit
    close(pts, g[f"{tag}_pts"], 4e-6)
    assert bool((valid == T(g[f"{tag}_valid"])).all())
    assert bool((labels == T(g[f"{tag}_labels"])).all())


def test_g7_get_training_samples(golden):
    g = golden("g7_sample")
    rgbs, depth, origins, dirs_w, _, _ = O.get_training_samples(
        T(g["gts_rgbs_batch"]), T(g["gts_depth_batch"]), T(g["gts_t_wc"]), T(g["gts_bbox"]),
        T(g["gts_kf_ids"]), T(g["gts_u_w"]), T(g["gts_u_h"]), T(g["gts_rays_dir_cache"]))
    assert bool((rgbs[..., :3] == T(g["gts_rgb"])).all())
    close(depth, g["gts_depth"], 0)
    z, pts, valid, labels = O.sample_3d_points(rgbs[..., 3], depth, origins, dirs_w, T(g["gts_u"]),
                                               T(g["gts_g"]), 1, 9, 0.1, 0.05)
    close(z, g["gts_z"], 1e-6)
    close(pts, g["gts_pts"], 4e-6)
    assert bool((labels == T(g["gts_labels"])).all())
    W, H, fx, fy, cx, cy = [float(x) for x in g["gts_cam"]]
    close(O.rays_dirs(int(W), int(H), fx, fy, cx, cy), g["gts_rays_dir_cache"], 0)


@pytest.mark.parametrize("tag", ["pd5", "pd3"])
def test_g7b_partfeat(golden, tag):
    """get_training_samples with part_mode on: the 7th output (vmap.py:437-452)."""
    g = golden("g7b_partfeat")
    pd, stride, Cf, W, H = [int(x) for x in g[f"{tag}_meta"]]
    rgbs, depth, _, _, idx_w, idx_h = O.get_training_samples(
        T(g[f"{tag}_rgbs_batch"]), T(g[f"{tag}_depth_batch"]), T(g[f"{tag}_t_wc"]), T(g[f"{tag}_bbox"]),
        T(g[f"{tag}_kf_ids"]), T(g[f"{tag}_u_w"]), T(g[f"{tag}_u_h"]), T(g["rays_dir_cache"]))
    assert bool((rgbs[..., :3] == T(g[f"{tag}_rgb"])).all())
    close(depth, g[f"{tag}_depth"], 0)
    pf = O.sample_partfeat(T(g[f"{tag}_global_partfeat"]), g[f"{tag}_use_frame"], stride, pd, T(g[f"{tag}_kf_ids"]),
                           idx_w, idx_h)
    assert pf.shape == g[f"{tag}_partfeat"].shape
    assert np.array_equal(pf.numpy(), g[f"{tag}_partfeat"])          # a gather: bit-exact


@pytest.mark.parametrize("tag", ["nofeat", "feat"])
def test_g15_forloop_first_step(golden, tag):
    """The reference's forloop strategy (fixture G15) equals the stacked oracle step: loss and gradients of the
    first iteration (the reference's two strategies agree to 1.9e-6)."""
    g = golden(f"g15_forloop_{tag}")
    K, R, n1, n2, feat_on = [int(x) for x in g["meta"]]
    rec = _run_oracle_steps(g, 1, bool(feat_on),
                            lambda it: synthetic.random_batch(K, R, n1, n2, seed=1500 + it, feat_dim=512))
    np.testing.assert_allclose(rec["loss"][0], g["loss"][0], rtol=2e-5)
    for i in range(19):
        gr = rec["grads"][0][i]
        if g["none_grad"][0][0][i]:
            assert gr is None
        else:
            scale = max(1e-3, float(np.abs(g[f"grad0_{i}"]).max()))
            close(gr, g[f"grad0_{i}"], 2e-5 * scale)
        close(rec["params"][0][i], g[f"param0_{i}"], 2e-6)


def test_g8_box(golden):
    g = golden("g8_box")
    near, far, hit = O.ray_box_intersection(T(g["o"]), T(g["d"]), T(g["bmin"]), T(g["bmax"]))
    close(near, g["near"], 0)
    close(far, g["far"], 0)
    assert bool((hit == T(g["hit"])).all())
    ow, dw = O.origin_dirs_W(T(g["T"]), T(g["dc"]))
    close(ow, g["ow"], 0)
    close(dw, g["dw"], 1e-6)
    ow2, dw2 = O.origin_dirs_W(T(g["T"]), T(g["dc2"]))
    close(dw2, g["dw2"], 1e-6)


@pytest.mark.parametrize("tag", ["nofeat", "feat"])
def test_g10_bg(golden, tag):
    g = golden(f"g10_bg_{tag}")
    K, R, N, M, feat_on, H = [int(x) for x in g["meta"]]
    g = dict(g)
    g["scale"] = np.full(K, 5.0, np.float32)
    rec = _run_oracle_steps(g, 1, bool(feat_on),
                            lambda it: synthetic.random_batch(K, R, N, M, seed=1000 + it, feat_dim=512))
    np.testing.assert_allclose(rec["loss"], g["loss"], rtol=2e-5)
    for i in range(19):
        if rec["grads"][0][i] is None:
            continue
        scale = max(1e-3, float(np.abs(g[f"grad0_{i}"]).max()))
        close(rec["grads"][0][i], g[f"grad0_{i}"], 3e-5 * scale)


def render_eval_psnr(fc, B, scale, ev):
    with torch.no_grad():
        out = O.render_forward(fc, B, scale, T(ev["pts"]), T(ev["z"]), with_feat=False)
    return O.psnr(out["rgb"], T(ev["gt_rgb"])), out


@pytest.mark.parametrize("tag", ["nofeat", "feat"])
def test_g9_psnr_scene(golden, tag):
    """300 iterations on the analytic ellipsoid scene: the oracle reproduces the reference's loss
    curve and its PSNR on held-out rays."""
    g = dict(golden(f"g9_psnr_{tag}"))
    K, R, N, M, steps, eval_R, eval_S, scene_seed = [int(x) for x in g["meta"]]
    scene = synthetic.EllipsoidScene.make(K, 512, seed=scene_seed)
    rec = _run_oracle_steps(g, steps, tag == "feat",
                            lambda it: scene.batch(R, N, M, seed=9000 + it, with_feat=True))
    np.testing.assert_allclose(rec["loss"][:20], g["loss"][:20], rtol=1e-4)
    np.testing.assert_allclose(rec["loss"][-1], g["loss"][-1], rtol=2e-2)
    ev = scene.eval_rays(eval_R, eval_S)
    fc, B = rec["params"][-1][:18], rec["params"][-1][18]
    p, out = render_eval_psnr(fc, B, torch.full((K,), 2.0), ev)
    assert abs(p - float(g["psnr"])) < 0.1, (p, float(g["psnr"]))


@pytest.mark.parametrize("H", [32, 128])
def test_g11_render_2d_syn(golden, H):
    """Trainer.sample_points_bbox + sceneObject.render_2D_syn (trainer.py:130-198, vmap.py:604-685)."""
    g = golden("g11_render")
    tag = f"h{H}"
    fc = [T(g[f"{tag}_p{i}"]) for i in range(18)]
    r = O.render_2d_syn(fc, T(g[f"{tag}_B"]), float(g["scale"]), T(g["T_WC"]), T(g["rays_dir"]), g["box_center"],
                        g["box_R"], g["box_extent"], T(g["mask_in"]), T(g[f"{tag}_u"]))
    assert r is not None
    np.testing.assert_allclose(r["z_vals"], g[f"{tag}_z_vals"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(r["pts"], g[f"{tag}_pts"], rtol=0, atol=2e-6)
    assert np.array_equal(r["mask"].numpy(), g[f"{tag}_mask_out"])
    np.testing.assert_allclose(r["depth"], g[f"{tag}_depth"], rtol=1e-5, atol=1e-6)
    assert np.abs(r["color"].numpy().astype(int) - g[f"{tag}_color"].astype(int)).max() <= 1
    np.testing.assert_allclose(r["feat"].detach(), g[f"{tag}_feat"], rtol=1e-4, atol=1e-5)


def test_g9b_oracle_reproduces_reference_psnr50():
    """G9b (the PSNR scene of SURVEY.md 8(d), tests/golden/make_g9b_ensemble.py): the oracle, started from the
    reference's initial weights for the first fixture seed (the host mirror of trainer.py:36-44 draws them), reproduces
    the reference's PSNR after 50 iterations -- the fixture the GPU ensembles are held to is what the reference
    computes, and the oracle agrees with it."""
    from openobj_amd import psnr_scene
    ref = psnr_scene.reference_ensemble_b(False)
    assert ref is not None and len(ref["seeds"]) >= 300
    s = psnr_scene.G9B
    assert [int(x) for x in ref["meta"][:8]] == [s["K"], s["R"], s["N"], s["M"], s["steps"], s["early"], s["eval_R"], s["eval_S"]]
    er = psnr_scene.EnsembleRun("cpu", with_feat=False)
    seed = int(ref["seeds"][0])
    arena = er.initial_arena([seed])
    params = [p.clone().requires_grad_(True) for p in arena.views()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    scale = torch.full((s["K"],), float(er.cfg.obj_scale))
    for it in range(s["early"]):
        b = er.batches[it]
        loss, _ = O.train_forward_loss(params[:18], params[18], scale, b["pts"], b["gt_depth"], b["gt_rgb"], b["labels"], b["z"])
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                if g is not None:
                    O.adamw_step(p, g, mm, vv, it + 1, er.cfg.learning_rate, er.cfg.weight_decay)
    with torch.no_grad():
        out = O.render_forward([p.detach() for p in params[:18]], params[18].detach(), scale, er.ev["pts"], er.ev["z"],
                               with_feat=False)
    p50 = O.psnr(out["rgb"], er.ev["gt_rgb"])
    assert abs(p50 - float(ref["psnr50"][0])) < 0.02, (p50, float(ref["psnr50"][0]))
