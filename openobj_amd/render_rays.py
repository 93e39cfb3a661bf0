"""render_rays -- the reference's compositing helpers (render_rays.py:6-146) on the HIP kernels.

occupancy_activation / occupancy_to_termination / render run objnerf_occupancy / objnerf_composite / objnerf_render;
render_loss / reduce_batch_loss / make_3D_grid run the small kernels of objnerf_helpers.hip.  The fused
iteration uses none of them (its loss lives inside the training kernels and objnerf_step_batch_loss); they
are provided for callers of the reference API and operate on device tensors.
"""
import torch

from . import ops


def occupancy_activation(alpha, distances=None):
    if distances is not None:                               # render_rays.py:10-11 (never used by the reference)
        return 1.0 - torch.exp(-alpha * distances)
    return ops.occupancy(alpha.contiguous())                # render_rays.py:13


def occupancy_to_termination(occupancy, is_batch=False):
    """render_rays.py:32-54: w_i = occ_i * prod_{j<i} (1 - occ_j + 1e-10)."""
    S = occupancy.shape[-1]
    out = ops.composite(occupancy.reshape(-1, S).contiguous(), None, None, want_term=True,
                        input_is_occupancy=True)
    return out["term"].reshape(occupancy.shape)


def render(termination, vals, dim=-1):
    """Weighted sum of per-sample values along a ray (reference render_rays.py:56-63; the reference multiplies tensors
    of equal shape and sums over `dim`): objnerf_render.  Two call forms:
      render(term [..., S], vals [..., S])            dim = -1  (loss.py:28 depth, vmap.py:664)
      render(term [..., S, 1], vals [..., S, C], -2)  the reference's vector form (loss.py:34,82, vmap.py:670,678)
    and, beyond the reference, vals with a trailing channel axis next to an un-expanded termination [..., S]."""
    if dim in (-2, termination.dim() - 2) and termination.dim() >= 2 and termination.shape[-1] == 1 \
            and vals.dim() == termination.dim():
        return ops.render(termination[..., 0].contiguous(), vals.contiguous())
    if dim not in (-1, termination.dim() - 1):
        raise ValueError("render: the sample axis must be termination's last axis (or dim=-2 with a trailing "
                         "singleton axis on termination, the reference's vector form)")
    return ops.render(termination.contiguous(), vals.expand_as(termination).contiguous()
                      if vals.dim() == termination.dim() else vals.contiguous())


_LOSS_MODES = {"L1": 0, "L2": 1, "cos": 2}


def render_loss(render, gt, loss="L1", normalise=False):
    """Per-element residual of a rendered quantity against its target (reference render_rays.py:65-83): "L1"
    absolute, "L2" squared, "cos" one minus the cosine similarity over the last axis; objnerf_render_loss."""
    if loss not in _LOSS_MODES:
        raise ValueError("loss type {} not implemented!".format(loss))
    return ops.render_loss(render, gt, _LOSS_MODES[loss], normalise)


class LossExplode(RuntimeError):
    """The reference prints 'loss explode' and exit(-1)s (render_rays.py:109-111)."""


STATUS_EXPLODE = 1       # a loss term above 1e5: the reference's test (render_rays.py:109-111) -> LossExplode
STATUS_NONFINITE = 2     # a NaN / Inf loss term: `nan > 100000` is False, the reference carries on -> a warning, once


def check_status(status) -> int:
    """The ONE contract for the kernels' status word, whichever path wrote it (fused hidden-32, small-batch, layer-wise,
    hidden-256, objnerf_reduce_batch_loss): bit 0 raises LossExplode where the reference exits; bit 1 (a non-finite
    term) is surfaced as a RuntimeWarning and the run carries on, as the reference does.  Returns the word."""
    s = int(status.item()) if hasattr(status, "item") else int(status)
    if s & STATUS_EXPLODE:
        raise LossExplode("loss explode")
    if s & STATUS_NONFINITE:
        import warnings
        warnings.warn("objnerf: a loss term is NaN / Inf; carrying on as the reference does (render_rays.py:109-111 "
                      "tests only `> 100000`)", RuntimeWarning, stacklevel=2)
    return s


def reduce_batch_loss(loss_mat, var=None, avg=True, mask=None, loss_type="L1"):
    """Masked, optionally information-weighted reduction of a [K, R] loss matrix to one value per object
    (reference render_rays.py:85-117) on objnerf_reduce_batch_loss: if ANY object's mask is empty the result is zero
    for ALL objects (the reference's early return), a mean above 1e5 raises LossExplode where the reference exits."""
    if mask is None:
        raise ValueError("reduce_batch_loss needs a mask (the reference sums it unconditionally, render_rays.py:88)")
    out, status = ops.reduce_batch_loss(loss_mat, var, mask, l2=(loss_type == "L2"), avg=bool(avg))
    if avg:
        check_status(status)
    return out


def make_3D_grid(occ_range=[-1., 1.], dim=256, device="cuda:0", transform=None, scale=None):
    """[dim, dim, dim, 3] lattice over occ_range^3, scaled per axis and moved into the world by `transform` [4,4]
    (reference render_rays.py:119-146); one objnerf_make_grid launch."""
    return ops.make_grid(int(dim), float(occ_range[0]), float(occ_range[1]), scale, transform, device)
