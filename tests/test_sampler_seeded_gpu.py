"""The seeded form of the sampler (objnerf_sample_rays with u_w = u_h = u = g = NULL; reference vmap.py:386-554,
utils.py:342-397).  The reference draws with torch.rand / normal_, whose stream cannot be replayed, so parity for
seeded draws has two legs:
  * EXACT: the seeded kernels equal the injected-draw kernels (pinned to the reference by fixture G7) when the injected
    numbers are the same Philox4x32-10 outputs, recomputed here in numpy from the published algorithm;
  * DISTRIBUTIONAL: Kolmogorov-Smirnov tests of the draws recovered from the outputs against the analytic laws and
    against the draws the reference's own generator produced (recorded in G7: metric_u, metric_g).
"""
import numpy as np
import pytest
import torch
from scipy import stats

from conftest import T
from openobj_amd import cfg as ocfg
from openobj_amd import ops
from openobj_amd import vmap as ovmap

pytestmark = pytest.mark.gpu

S_KEYFRAME, S_PIXEL, S_BINS_U, S_BINS_G = 1, 2, 4, 5       # objnerf_philox.h Stream


def philox4x32_10(ctr, k0, k1):
    """ctr: uint32 [..., 4]; Salmon et al. 2011, 10 rounds."""
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    x, y, z, w = [ctr[..., i].astype(np.uint64) for i in range(4)]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = np.uint64(M0) * x, np.uint64(M1) * z
        x, y, z, w = ((p1 >> np.uint64(32)) ^ y ^ k0) & mask, p1 & mask, ((p0 >> np.uint64(32)) ^ w ^ k1) & mask, p0 & mask
        k0, k1 = (k0 + np.uint64(W0)) & mask, (k1 + np.uint64(W1)) & mask
    return np.stack([x, y, z, w], axis=-1).astype(np.uint32)


def uniform4(seed, stream, a, b, c):
    a, b, c = np.broadcast_arrays(np.asarray(a, np.uint32), np.asarray(b, np.uint32), np.asarray(c, np.uint32))
    ctr = np.stack([a, b, c, np.full(a.shape, stream, np.uint32)], axis=-1)
    r = philox4x32_10(ctr, seed & 0xFFFFFFFF, seed >> 32)
    return (r >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)


def philox_draws(seed, draw, obj, nk, tail, n_frames, n_px, N, M, eps):
    """The numbers the seeded kernels use, in the layout of the injected arguments."""
    n, Sn = n_frames * n_px, N + M
    tag = draw << 3
    f = np.arange(n_frames)
    uk = uniform4(seed, S_KEYFRAME | tag, obj, 0, f >> 2)[f, f & 3]
    kf = np.minimum((uk * np.float32(nk)).astype(np.int64), nk - 1)
    for t in range(2):
        if tail[t] >= 0:
            kf[n_frames - 2 + t] = tail[t]
    i = np.arange(n)
    px = uniform4(seed, S_PIXEL | tag, obj, i, 0)
    s = np.arange(Sn)
    u = uniform4(seed, S_BINS_U | tag, obj, i[:, None], (s >> 2)[None, :])[i[:, None], s[None, :], (s & 3)[None, :]]
    m = np.arange(M)
    r = uniform4(seed, S_BINS_G | tag, obj, i[:, None], (m >> 1)[None, :])
    u1 = 1.0 - r[i[:, None], m[None, :], (2 * (m & 1))[None, :]].astype(np.float64)
    u2 = r[i[:, None], m[None, :], (2 * (m & 1) + 1)[None, :]].astype(np.float64)
    g = (np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2) * (eps / 3.0)).astype(np.float32)
    return dict(kf_ids=kf, u_w=px[:, 0].reshape(n_frames, n_px), u_h=px[:, 1].reshape(n_frames, n_px), u=u, g=g)


def g7_object(golden, dev, N=1, M=9, part_mode=0):
    g = golden("g7_sample")
    W, H = g["gts_rgbs_batch"].shape[1:3]
    c = ocfg.Config(ocfg.replica_room0_config(train_device=str(dev), **{"model.keyframe_buffer_size": 4,
                                                                         "trainer.part_mode": part_mode}))
    c.obj_id = 1
    c.W, c.H = int(W), int(H)
    c.fx = c.fy = 30.0
    c.cx, c.cy = 19.5, 14.5
    cam = ovmap.cameraInfo(c)
    rb = T(g["gts_rgbs_batch"])
    obj = ovmap.sceneObject(c, 1, rb[0, :, :, :3].to(dev), T(g["gts_depth_batch"])[0].to(dev), rb[0, :, :, 3].to(dev),
                            T(g["gts_bbox"])[0].to(dev), T(g["gts_t_wc"])[0].to(dev), 0)
    obj.rgbs_batch.copy_(rb.to(dev))
    obj.depth_batch.copy_(T(g["gts_depth_batch"]).to(dev))
    obj.t_wc_batch.copy_(T(g["gts_t_wc"]).to(dev))
    obj.bbox.copy_(T(g["gts_bbox"]).to(dev))
    obj.n_keyframes = 4
    obj.lastest_kf_queue = [2, 3]
    obj.n_bins_cam2surface, obj.n_bins = N, M
    return g, cam, obj


@pytest.mark.parametrize("N,M", [(1, 9), (16, 48)])
def test_seeded_equals_injected_philox(golden, dev, N, M):
    """Same numbers, two routes: generated inside the kernels vs injected (the route G7 pins to the reference)."""
    _, cam, obj = g7_object(golden, dev, N, M)
    seed, draw, nf, npx = 0x1234567890ABCDEF, 77, 9, 33
    meta = torch.tensor(obj.kf_meta(), dtype=torch.int32, device=dev)
    o = ops.sample_rays_seeded(obj.keyframe_store(), 4, obj.frames_width, obj.frames_height, cam.rays_dir_cache, meta, nf,
                               npx, N, M, obj.surface_eps, obj.stop_eps, float(obj.min_bound), float(obj.obj_center),
                               seed=seed, draw=draw, want_pts=True, record=True)
    d = philox_draws(seed, draw, obj.obj_id, 4, [2, 3], nf, npx, N, M, obj.surface_eps)
    assert np.array_equal(o["kf"].cpu().numpy(), d["kf_ids"])
    dd = {k: T(v).to(dev) for k, v in d.items()}
    rgb, depth, valid, labels, pts, z = ops.sample_rays(obj.rgbs_batch, obj.depth_batch, obj.t_wc_batch, obj.bbox,
                                                        cam.rays_dir_cache, dd["kf_ids"], dd["u_w"], dd["u_h"], dd["u"],
                                                        dd["g"], N, M, obj.surface_eps, obj.stop_eps,
                                                        float(obj.min_bound), float(obj.obj_center))
    assert torch.equal(o["rgb"].reshape(-1, 3), rgb.reshape(-1, 3)) and torch.equal(o["labels"], labels)
    assert torch.equal(o["valid"], valid) and torch.equal(o["depth"].reshape(-1), depth.reshape(-1))
    z, zs = z.reshape(-1, N + M), o["z"].reshape(-1, N + M)
    normal = (valid & (labels == 1)).cpu()
    assert int(normal.sum()) > 20 and int((~normal).sum()) > 20
    assert torch.equal(zs[~normal], z[~normal])                       # uniform placements: the same bits
    assert torch.equal(zs[normal][:, :N], z[normal][:, :N])
    # Box-Muller on the device (logf / cosf) vs numpy float64: the same normals up to the functions' rounding
    assert (zs[normal] - z[normal]).abs().max().item() < 2e-6
    assert (o["pts"].reshape(-1, 3) - pts.reshape(-1, 3)).abs().max().item() < 4e-6
    # pixel record = the truncated float index of the injected route
    bb = obj.bbox[dd["kf_ids"]][:, None, :]
    iw = (dd["u_w"] * (bb[..., 1] - bb[..., 0]) + bb[..., 0]).long().reshape(-1)
    ih = (dd["u_h"] * (bb[..., 3] - bb[..., 2]) + bb[..., 2]).long().reshape(-1)
    assert torch.equal(o["px"][:, 0].long(), iw) and torch.equal(o["px"][:, 1].long(), ih)


@pytest.mark.parametrize("pd,stride,C", [(5, 1, 512), (3, 2, 20)])
def test_seeded_partfeat_equals_injected_philox(golden, dev, pd, stride, C):
    """The part-feature gather (vmap.py:437-452) of the SEEDED sampler equals that of the injected sampler -- which
    fixture G7b pins to the reference bit for bit (test_get_training_samples_partfeat_g7b) -- when the injected draws
    are the same Philox outputs; also through sceneObject.get_training_samples and StackedSampler (part_mode on)."""
    _, cam, obj = g7_object(golden, dev, part_mode=1)
    obj.part_down, obj.stride = pd, stride
    obj.use_frame = np.array([0., 2., 1., 3.]) * stride
    W, H = obj.frames_width, obj.frames_height
    gpf = torch.randn(4, -(-W // pd), -(-H // pd), C, generator=torch.Generator().manual_seed(1)).to(dev)
    seed, draw, nf, npx = 99, 5, 9, 300
    meta = torch.tensor(obj.kf_meta(), dtype=torch.int32, device=dev)
    pfa = (gpf, obj.use_frame, stride, pd)
    o = ops.sample_rays_seeded(obj.keyframe_store(), 4, W, H, cam.rays_dir_cache, meta, nf, npx, 1, 9, obj.surface_eps,
                               obj.stop_eps, float(obj.min_bound), float(obj.obj_center), seed=seed, draw=draw,
                               want_pts=True, partfeat=pfa)
    d = philox_draws(seed, draw, obj.obj_id, 4, [2, 3], nf, npx, 1, 9, obj.surface_eps)
    dd = {k: T(v).to(dev) for k, v in d.items()}
    r = ops.sample_rays(obj.rgbs_batch, obj.depth_batch, obj.t_wc_batch, obj.bbox, cam.rays_dir_cache, dd["kf_ids"],
                        dd["u_w"], dd["u_h"], dd["u"], dd["g"], 1, 9, obj.surface_eps, obj.stop_eps,
                        float(obj.min_bound), float(obj.obj_center), partfeat=pfa)
    assert o["partfeat"].shape == (nf * npx, C)
    assert torch.equal(o["partfeat"], r[6].reshape(-1, C))
    # the same gather restated with torch indexing from the injected draws (the reference's three lines)
    bb = obj.bbox[dd["kf_ids"]][:, None, :]
    iw = dd["u_w"] * (bb[..., 1] - bb[..., 0]) + bb[..., 0]
    ih = dd["u_h"] * (bb[..., 3] - bb[..., 2]) + bb[..., 2]
    fid = (torch.tensor(obj.use_frame).to(dev)[dd["kf_ids"]][:, None] / stride).long()
    ref = gpf[fid, torch.floor(iw / pd).long(), torch.floor(ih / pd).long()]
    assert torch.equal(r[6], ref)
    # the mirrored call surface
    out = obj.get_training_samples(nf, npx, cam.rays_dir_cache, gpf, draws=dd)
    assert len(out) == 7 and torch.equal(out[6], ref)
    out = obj.get_training_samples(nf, npx, cam.rays_dir_cache, gpf, seed=seed)
    assert out[6].shape == (nf, npx, C)
    ss = ovmap.StackedSampler([obj, obj])
    so = ss.sample(nf, npx, cam.rays_dir_cache, gpf, draws={k: torch.stack([v, v]) for k, v in dd.items()})
    assert torch.equal(so[6][0], ref.reshape(-1, C)) and torch.equal(so[6][1], ref.reshape(-1, C))
    so = ss.sample(nf, npx, cam.rays_dir_cache, gpf, seed=seed, compact=True)
    assert so[6].shape == (2, nf * npx, C)
    assert obj.get_training_samples(nf, npx, cam.rays_dir_cache, None, draws=dd)[6] is None


def test_seeded_is_reproducible_and_keyed(golden, dev):
    _, cam, obj = g7_object(golden, dev)
    kw = dict(seed=5)
    def run(**k):
        meta = torch.tensor(obj.kf_meta(), dtype=torch.int32, device=dev)
        if "obj" in k:
            meta[3] = k.pop("obj")
        return ops.sample_rays_seeded(obj.keyframe_store(), 4, obj.frames_width, obj.frames_height, cam.rays_dir_cache,
                                      meta, 6, 16, 1, 9, obj.surface_eps, obj.stop_eps, 0.0, 0.0, want_pts=True, **k)
    a, b = run(seed=5, draw=3), run(seed=5, draw=3)
    assert all(torch.equal(a[k], b[k]) for k in ("rgb", "depth", "labels", "z", "pts"))
    for other in (run(seed=6, draw=3), run(seed=5, draw=4), run(seed=5, draw=3, obj=2)):
        assert not torch.equal(a["z"], other["z"])
    # the default draw counter advances by itself: two calls, two batches
    assert not torch.equal(run(seed=5)["z"], run(seed=5)["z"])


def test_compact_form_trains_like_the_point_form(golden, dev):
    """origins / dirs / z instead of pts: the same points (two roundings, vmap.py:548-551), bit-identical iteration on
    the fused kernel, and the layer-wise path accepts the form too."""
    from openobj_amd import init as obj_init
    _, cam, obj = g7_object(golden, dev, 16, 48)
    meta = torch.tensor([obj.kf_meta()] * 3, dtype=torch.int32, device=dev)
    meta[:, 3] = torch.arange(3)
    table = ops.keyframe_table([obj.keyframe_store()] * 3)
    kw = dict(seed=11, draw=1)
    args = (table, 4, obj.frames_width, obj.frames_height, cam.rays_dir_cache, meta, 8, 16, 16, 48, obj.surface_eps,
            obj.stop_eps, 0.0, 0.0)
    full = ops.sample_rays_seeded(*args, want_pts=True, **kw)
    comp = ops.sample_rays_seeded(*args, want_pts=False, **kw)
    assert comp["pts"] is None and full["origins"] is None
    assert torch.equal(full["z"], comp["z"]) and torch.equal(full["rgb"], comp["rgb"])
    assert not torch.equal(comp["z"][0], comp["z"][1])                 # objects have their own streams
    pts = (comp["origins"][:, :, None, :] + comp["dirs"][:, :, None, :] * comp["z"][..., None]) - 0.0
    assert torch.equal(pts, full["pts"])
    K, R, S = comp["z"].shape
    arena = ops.ParamArena(K, ops.NetShape(32, 512, 6), dev)
    arena.load_stacked(obj_init.init_stacked(K, 32, 512, seed=3))
    common = {"z": comp["z"], "gt_depth": comp["depth"], "gt_rgb": comp["rgb"].float() / 255.0, "labels": comp["labels"]}
    outs = []
    for form in ({"pts": full["pts"]}, {"origins": comp["origins"], "dirs": comp["dirs"]}):
        for layerwise in (False, True):
            ws = ops.TrainWorkspace(arena, K, R, S, False, layerwise=layerwise)
            ops.train_step(arena, ws, dict(common, **form), with_feat=False, layerwise=layerwise)
            outs.append((ws.loss_terms.clone(), ws.grads.clone()))
    assert torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])        # fused: same bits
    scale = outs[1][1].abs().max().item()
    assert (outs[1][1] - outs[3][1]).abs().max().item() <= 2e-6 * scale                     # (split-K atomics order)
    assert ((outs[1][0] - outs[3][0]).abs() <= 1e-6 * outs[1][0].abs().clamp(min=1.0)).all()     # (atomic sums)


def _recover_uniforms(z, lo, hi, n):
    """Invert stratified_bins: z[s] = lo + (hi - lo) (s + u) / n."""
    s = torch.arange(n, dtype=torch.float64)
    return ((z.double() - lo[:, None].double()) / (hi - lo)[:, None].double() * n - s).clamp(0.0, 1.0)


def test_seeded_draws_follow_the_reference_distributions(golden, dev):
    """KS tests at the 0.1 % level (fixed seed: deterministic).  Uniform bins, keyframe and pixel choice, and the
    sorted clipped normals around the surface -- against the analytic laws and the reference generator's own
    recorded draws (G7 metric_u / metric_g, torch.rand / normal_ of the reference run)."""
    g, cam, obj = g7_object(golden, dev, 16, 48)
    eps, N, M = obj.surface_eps, 16, 48
    meta = torch.tensor(obj.kf_meta(), dtype=torch.int32, device=dev)
    nf, npx = 64, 64
    o = ops.sample_rays_seeded(obj.keyframe_store(), 4, obj.frames_width, obj.frames_height, cam.rays_dir_cache, meta, nf,
                               npx, N, M, eps, obj.stop_eps, 0.0, 0.0, seed=2024, draw=1, want_pts=False, record=True)
    z, d = o["z"].cpu(), o["depth"].cpu()
    valid, labels = o["valid"].cpu(), o["labels"].cpu()
    # (the fixture stores the reference's draws where it drew them and 0 elsewhere: label-dependent branches)
    ref_u = np.concatenate([g[k].reshape(-1) for k in ("metric_u", "obj_u", "bg_u", "gts_u")])
    ref_g = np.concatenate([g[k].reshape(-1) for k in ("metric_g", "obj_g", "bg_g", "gts_g")])
    ref_u, ref_g = ref_u[ref_u != 0], ref_g[ref_g != 0]
    assert len(ref_u) > 2000 and len(ref_g) > 1000
    # 1. uniforms of the camera-to-surface bins of every valid ray
    u1 = _recover_uniforms(z[valid][:, :N], torch.zeros(int(valid.sum())), d[valid] - eps, N).reshape(-1).numpy()
    assert len(u1) > 20000
    assert stats.kstest(u1, "uniform").pvalue > 1e-3
    assert stats.ks_2samp(u1, ref_u).pvalue > 1e-3
    # 2. uniforms of the near-surface bins of valid rays off the object
    off = valid & (labels != 1)
    u2 = _recover_uniforms(z[off][:, N:], d[off] - eps, d[off] + obj.stop_eps, M).reshape(-1).numpy()
    assert len(u2) > 5000 and stats.kstest(u2, "uniform").pvalue > 1e-3
    # 3. normals on the object: pooled (sorting within a ray does not change the pooled sample), clipped at +-eps = 3 sigma
    on = valid & (labels == 1)
    gz = (z[on][:, N:] - d[on][:, None]).double()
    assert bool((gz[:, 1:] >= gz[:, :-1]).all())                          # sorted along the ray (utils.py:393)
    assert float(gz.abs().max()) <= eps + 1e-6
    gz = gz.reshape(-1).numpy()
    assert len(gz) > 5000
    clipped = lambda x: np.clip(x, -eps, eps)
    assert stats.ks_2samp(gz, clipped(ref_g)).pvalue > 1e-3               # the reference's normal_ draws
    assert stats.kstest(gz / (eps / 3.0), lambda x: np.where(x >= 3.0, 1.0, np.where(x < -3.0, 0.0, stats.norm.cdf(x)))
                        ).pvalue > 1e-3
    # 4. keyframes: the last two of the draw are the latest two slots, the others uniform over the stored keyframes
    kf = o["kf"].cpu().numpy()
    assert list(kf[-2:]) == [2, 3]
    counts = np.bincount(kf[:-2], minlength=4)
    assert stats.chisquare(counts).pvalue > 1e-3
    # 5. pixels: inside the keyframe's box, uniform over it
    bb = obj.bbox.cpu().numpy()[kf].repeat(npx, axis=0)
    px = o["px"].cpu().numpy().astype(np.float64)
    assert bool((px[:, 0] >= np.floor(bb[:, 0])).all() and (px[:, 0] <= bb[:, 1]).all())
    assert bool((px[:, 1] >= np.floor(bb[:, 2])).all() and (px[:, 1] <= bb[:, 3]).all())
    wide = (bb[:, 1] - bb[:, 0]) >= 8
    fw = (px[wide, 0] + 0.5 - bb[wide, 0]) / (bb[wide, 1] - bb[wide, 0])
    assert stats.kstest(fw, "uniform").statistic < 0.5 / 8 + 0.03         # (discretised to pixels)


def test_scene_object_and_stacked_sampler_default_to_seeded_draws(golden, dev):
    """The reference API with no injected draws: no random tensor, reference tuple layout, part features looked up at
    the drawn pixels."""
    g, cam, obj = g7_object(golden, dev, 1, 9, part_mode=1)
    obj.part_mode, obj.part_down, obj.stride = True, 2, 1
    obj.use_frame = [0, 1, 2, 3]
    W, H = obj.frames_width, obj.frames_height
    gp = torch.arange(4 * (W // 2) * (H // 2), dtype=torch.float32, device=dev).reshape(4, W // 2, H // 2, 1).repeat(1, 1, 1, 3)
    rgb, depth, valid, labels, pts, z, pf = obj.get_training_samples(7, 5, cam.rays_dir_cache, gp, seed=9)
    assert rgb.shape == (7, 5, 3) and depth.shape == (7, 5) and pts.shape == (7, 5, 10, 3) and z.shape == (7, 5, 10)
    assert pf.shape == (7, 5, 3) and valid.shape == (35,) and labels.shape == (35,)
    # the looked-up feature encodes (frame, w / 2, h / 2): it must be the pixel the colour came from
    code = pf.reshape(-1, 3)[:, 0].long()
    f, rem = code // ((W // 2) * (H // 2)), code % ((W // 2) * (H // 2))
    cw, ch = rem // (H // 2), rem % (H // 2)
    hit = 0
    for i in range(35):
        blk = obj.rgbs_batch[f[i], 2 * cw[i]:2 * cw[i] + 2, 2 * ch[i]:2 * ch[i] + 2, :3].reshape(-1, 3)
        hit += int((blk == rgb.reshape(-1, 3)[i]).all(dim=1).any())
    assert hit == 35
    sampler = ovmap.StackedSampler([obj, obj])
    out = sampler.sample(7, 5, cam.rays_dir_cache, gp, seed=9, compact=True)
    (origins, dirs), zz = out[4], out[5]
    assert origins.shape == (2, 35, 3) and dirs.shape == (2, 35, 3) and zz.shape == (2, 35, 10) and out[6].shape == (2, 35, 3)


def test_box_sampler_seeded_draws(dev):
    """Trainer.sample_points_bbox's stratified bins (trainer.py:171-176) with the draw generated in the kernel: the
    mid-points obey the injected formula for SOME u in [0, 1), and those u are uniform."""
    n, nb = 4096, 20
    gen = torch.Generator().manual_seed(3)
    near = torch.rand(n, generator=gen).to(dev)
    far = near + 0.2 + torch.rand(n, generator=gen).to(dev)
    dirs = torch.nn.functional.normalize(torch.randn(n, 3, generator=gen), dim=1).to(dev)
    origin = torch.tensor([0.1, -0.2, 0.3])
    z, pts = ops.box_points(origin, dirs, near, far, None, nb, seed=17)
    z2, _ = ops.box_points(origin, dirs, near, far, None, nb, seed=17)
    assert z.shape == (n, nb - 1) and pts.shape == (n, nb - 1, 3) and not torch.equal(z, z2)     # per-call counter
    assert torch.equal(pts, origin.to(dev)[None, None, :] + dirs[:, None, :] * z[..., None])
    # z[s] = lo + (hi - lo) (s + (u_s + u_{s+1} + 1) / 2) / nb  ->  the sum of two uniforms (triangular law on [0, 2])
    s = torch.arange(nb - 1, dtype=torch.float64)
    t = (((z.double().cpu() - near.double().cpu()[:, None]) / (far - near).double().cpu()[:, None] * nb - s) * 2 - 1)
    t = t.reshape(-1).numpy()
    assert t.min() > -1e-4 and t.max() < 2 + 1e-4
    tri = lambda x: np.where(x < 1, 0.5 * np.clip(x, 0, 1) ** 2, 1 - 0.5 * (2 - np.clip(x, 1, 2)) ** 2)
    # (neighbouring mid-points share one uniform: thin to every second bin for independent samples)
    t_ind = t.reshape(n, nb - 1)[:, ::2].reshape(-1)
    assert stats.kstest(t_ind, tri).pvalue > 1e-3
