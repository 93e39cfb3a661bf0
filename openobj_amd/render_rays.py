"""render_rays -- the reference's compositing helpers (render_rays.py:6-146) on the HIP kernels.

occupancy_activation / occupancy_to_termination / render run objnerf_occupancy / objnerf_composite.
render_loss / reduce_batch_loss / make_3D_grid are elementwise glue the fused iteration does not use
(its loss lives inside train_fused_kernel and objnerf_step_batch_loss); they are provided for callers
of the reference API and operate on device tensors.
"""
import torch
import torch.nn.functional as F

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
    """render_rays.py:56-63: weighted sum over the sample axis."""
    return (termination * vals).sum(dim=dim)


def render_loss(render, gt, loss="L1", normalise=False):   # render_rays.py:65-83
    residual = render - gt
    if loss == "L2":
        loss_mat = residual ** 2
    elif loss == "L1":
        loss_mat = torch.abs(residual)
    elif loss == "cos":
        loss_mat = 1 - F.cosine_similarity(render, gt, dim=-1)
    else:
        raise ValueError("loss type {} not implemented!".format(loss))
    if normalise:
        loss_mat = loss_mat / gt
    return loss_mat


class LossExplode(RuntimeError):
    """The reference prints 'loss explode' and exit(-1)s (render_rays.py:109-111)."""


def reduce_batch_loss(loss_mat, var=None, avg=True, mask=None, loss_type="L1"):   # render_rays.py:85-117
    mask_num = torch.sum(mask, dim=-1)
    if (mask_num == 0).any():          # cross-object early return
        loss = torch.zeros_like(loss_mat)
        if avg:
            loss = torch.mean(loss, dim=-1)
        return loss
    if var is not None:
        eps = 1e-4
        information = 1.0 / (var + eps) if loss_type == "L2" else 1.0 / (torch.sqrt(var) + eps)
        loss_weighted = loss_mat * information
    else:
        loss_weighted = loss_mat
    if avg:
        if mask is not None:
            loss = torch.sum(loss_weighted, dim=-1) / (torch.sum(mask, dim=-1) + 1e-10)
            if (loss > 100000).any():
                raise LossExplode("loss explode")
        else:
            loss = torch.mean(loss_weighted, dim=-1).sum()
    else:
        loss = loss_weighted
    return loss


def make_3D_grid(occ_range=[-1., 1.], dim=256, device="cuda:0", transform=None, scale=None):   # :119-146
    t = torch.linspace(occ_range[0], occ_range[1], steps=dim, device=device)
    grid = torch.meshgrid(t, t, t, indexing="ij")
    grid_3d = torch.cat((grid[0][..., None], grid[1][..., None], grid[2][..., None]), dim=3)
    if scale is not None:
        grid_3d = grid_3d * scale
    if transform is not None:
        R1 = transform[None, None, None, 0, :3]
        R2 = transform[None, None, None, 1, :3]
        R3 = transform[None, None, None, 2, :3]
        grid1 = (R1 * grid_3d).sum(-1, keepdim=True)
        grid2 = (R2 * grid_3d).sum(-1, keepdim=True)
        grid3 = (R3 * grid_3d).sum(-1, keepdim=True)
        grid_3d = torch.cat([grid1, grid2, grid3], dim=-1)
        grid_3d = grid_3d + transform[None, None, None, :3, 3]
    return grid_3d
