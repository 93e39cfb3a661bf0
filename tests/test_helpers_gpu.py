"""GPU: the reference's helper functions (render_rays.py:65-146, utils.py:309-397) on the kernels of objnerf_helpers.hip,
against the oracle and the reference-generated fixture G8; distribution checks for the counter-based draws."""
import numpy as np
import pytest
import torch
from scipy import stats

from conftest import T
from oracle import objnerf_oracle as O
from openobj_amd import ops, render_rays, utils

pytestmark = pytest.mark.gpu


def maxerr(a, b):
    return (torch.as_tensor(a).double().cpu() - torch.as_tensor(b).double().cpu()).abs().max().item()


def test_render_loss_modes(dev):
    g = torch.Generator().manual_seed(1)
    a, b = torch.randn(3, 17, 5, generator=g), torch.randn(3, 17, 5, generator=g)
    for mode in ("L1", "L2", "cos"):
        got = render_rays.render_loss(a.to(dev), b.to(dev), loss=mode)
        want = O.render_loss(a, b, loss=mode)
        assert tuple(got.shape) == tuple(want.shape) and maxerr(got, want) < 1e-6, mode
    f = torch.randn(4, 9, 512, generator=g)
    gt = torch.nn.functional.normalize(torch.randn(4, 9, 512, generator=g), dim=-1)
    assert maxerr(render_rays.render_loss(f.to(dev), gt.to(dev), loss="cos"), O.render_loss(f, gt, loss="cos")) < 1e-6
    with pytest.raises(ValueError):
        render_rays.render_loss(a.to(dev), b.to(dev), loss="huber")


def test_reduce_batch_loss_against_oracle(dev):
    g = torch.Generator().manual_seed(2)
    K, R = 4, 133
    loss_mat = torch.rand(K, R, generator=g)
    var = torch.rand(K, R, generator=g) * 0.3
    mask = torch.rand(K, R, generator=g) < 0.6
    lm = loss_mat * mask
    for kw in (dict(var=var), dict(var=None), dict(var=var, loss_type="L2")):
        got = render_rays.reduce_batch_loss(lm.to(dev), var=None if kw["var"] is None else var.to(dev), mask=mask.to(dev),
                                            loss_type=kw.get("loss_type", "L1"))
        want = O.reduce_batch_loss(lm, var=kw["var"], mask=mask, loss_type=kw.get("loss_type", "L1"))
        assert maxerr(got, want) < 1e-5 * max(1.0, float(want.abs().max()))
    got = render_rays.reduce_batch_loss(lm.to(dev), var=var.to(dev), avg=False, mask=mask.to(dev))
    assert maxerr(got, O.reduce_batch_loss(lm, var=var, avg=False, mask=mask)) < 1e-5 * float(1.0 / 1e-4)
    # the cross-object early return: one empty mask zeroes every object (render_rays.py:89-94)
    m2 = mask.clone()
    m2[2] = False
    got = render_rays.reduce_batch_loss((loss_mat * m2).to(dev), var=var.to(dev), mask=m2.to(dev))
    assert float(got.abs().max()) == 0.0 and tuple(got.shape) == (K,)
    # "loss explode" (render_rays.py:109-111)
    with pytest.raises(render_rays.LossExplode):
        render_rays.reduce_batch_loss((loss_mat * 1e9 * mask).to(dev), mask=mask.to(dev))


def test_make_3d_grid_g8(golden, dev):
    g = golden("g8_box")
    grid = render_rays.make_3D_grid(occ_range=[-1., 1.], dim=5, device=dev, transform=T(g["T"])[0],
                                    scale=torch.tensor([0.4, 0.6, 0.8]))
    assert tuple(grid.shape) == (5, 5, 5, 3) and maxerr(grid, g["grid"]) < 1e-6
    plain = render_rays.make_3D_grid(occ_range=[-2., 1.], dim=7, device=dev)
    t = torch.linspace(-2., 1., 7)
    want = torch.stack(torch.meshgrid(t, t, t, indexing="ij"), -1)
    assert maxerr(plain, want) < 1e-6


def test_ray_box_and_origin_dirs_g8(golden, dev):
    g = golden("g8_box")
    near, far, hit = utils.ray_box_intersection(T(g["o"]).to(dev), T(g["d"]).to(dev), T(g["bmin"]).to(dev), T(g["bmax"]).to(dev))
    assert torch.equal(hit.cpu(), T(g["hit"]))
    assert maxerr(near, g["near"]) < 1e-5 and maxerr(far, g["far"]) < 1e-5
    ow, dw = utils.origin_dirs_W(T(g["T"]).to(dev), T(g["dc"]).to(dev))
    assert torch.equal(ow.cpu(), T(g["ow"])) and maxerr(dw, g["dw"]) < 1e-6
    ow2, dw2 = utils.origin_dirs_W(T(g["T"]).to(dev), T(g["dc2"]).to(dev))
    assert tuple(dw2.shape) == tuple(g["dw2"].shape) and maxerr(dw2, g["dw2"]) < 1e-6


def test_stratified_bins_injected_and_drawn(dev):
    g = torch.Generator().manual_seed(3)
    n_rays, n_bins = 500, 12
    lo, hi = torch.rand(n_rays, generator=g), 2.0 + torch.rand(n_rays, generator=g)
    u = torch.rand(n_rays, n_bins, generator=g)
    got = utils.stratified_bins(lo.to(dev), hi.to(dev), n_bins, n_rays, device=dev, u=u.to(dev))
    assert maxerr(got, O.stratified_bins(lo, hi, n_bins, n_rays, u)) < 1e-6
    got = utils.stratified_bins(0.5, hi.to(dev), n_bins, n_rays, device=dev, u=u.to(dev))
    assert maxerr(got, O.stratified_bins(0.5, hi, n_bins, n_rays, u)) < 1e-6
    # drawn on the device: every value inside its bin, jitter uniform (Kolmogorov-Smirnov), fresh draws per call
    z = ops.stratified_bins(0.0, 1.0, 64, 4000, dev, seed=3, draw=1).cpu().double()
    edges = torch.arange(65, dtype=torch.float64) / 64
    assert bool((z >= edges[:-1] - 1e-6).all()) and bool((z <= edges[1:] + 1e-6).all())
    frac = ((z - edges[:-1]) * 64).reshape(-1).numpy()
    assert stats.kstest(frac, "uniform").pvalue > 1e-3
    z2 = utils.stratified_bins(0.0, 1.0, 64, 4000, device=dev).cpu().double()
    assert float((z2 - z).abs().max()) > 1e-3


def test_normal_bins_injected_and_drawn(dev):
    g = torch.Generator().manual_seed(4)
    n_rays, n_bins, delta = 300, 9, 0.1
    depth = 1.0 + torch.rand(n_rays, generator=g)
    gn = torch.randn(n_rays, n_bins, generator=g) * (delta / 3)
    got = utils.normal_bins_sampling(depth.to(dev), n_bins, n_rays, delta, device=dev, g=gn.to(dev))
    assert maxerr(got, O.normal_bins_sampling(depth, n_bins, n_rays, delta, gn)) < 1e-6
    # (fixed seed and call counter: the statistics below are of ONE known sample, not of whatever ran before)
    z = ops.normal_bins(torch.zeros(6000, device=dev), 48, delta, seed=4, draw=1).cpu().double()
    assert utils.normal_bins_sampling(torch.zeros(8, device=dev), 48, 8, delta, device=dev).shape == (8, 48)
    assert bool((z[:, 1:] >= z[:, :-1]).all()) and float(z.abs().max()) <= delta + 1e-7
    inner = z.reshape(-1).numpy()
    inner = inner[np.abs(inner) < delta * 0.999]                       # (clipping moves 0.27 % of the mass to +-delta)
    se = delta / 3 / np.sqrt(len(inner))
    assert abs(inner.std() - delta / 3) < 0.02 * delta / 3 and abs(inner.mean()) < 4 * se
    assert stats.kstest(inner[::7] / (delta / 3), stats.truncnorm(-3, 3).cdf).pvalue > 1e-3
