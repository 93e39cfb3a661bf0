"""loss.step_batch_loss (loss.py:5-103) on objnerf_step_batch_loss, differentiable w.r.t. alpha, color
and pred_partfeat like the reference's autograd graph (var detached, early return, scalings)."""
import torch

from . import ops
from .render_rays import LossExplode, check_status  # noqa: F401


class _StepBatchLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alpha, color, pred_partfeat, gt_depth, gt_color, sem_labels, z_vals, cs, os_, fs, gt_partfeat):
        out = ops.step_batch_loss(alpha.squeeze(-1).contiguous(), color, gt_depth, gt_color, sem_labels, z_vals,
                                  color_scaling=cs, opacity_scaling=os_, feat_scaling=fs, gt_feat=gt_partfeat,
                                  pred_feat=pred_partfeat, want_grads=True)
        check_status(out["status"])                 # render_rays.py:109-111 prints and exit(-1)s (bit 0 only)
        ctx.save_for_backward(out["d_alpha"], out["d_color"],
                              out["d_pred_feat"] if out["d_pred_feat"] is not None else torch.empty(0))
        ctx.alpha_shape = alpha.shape
        return out["total"].reshape(())

    @staticmethod
    def backward(ctx, g):
        d_alpha, d_color, d_pf = ctx.saved_tensors
        return (g * d_alpha.reshape(ctx.alpha_shape), g * d_color, (g * d_pf) if d_pf.numel() else None,
                None, None, None, None, None, None, None, None)


def step_batch_loss(alpha, color, gt_depth, gt_color, sem_labels, mask_depth, z_vals, color_scaling=5.0,
                    opacity_scaling=10.0, gt_partfeat=None, pred_partfeat=None, partfeat_scaling=5.0):
    """Same signature and return value `(loss, None)` as the reference.  `mask_depth` is accepted and
    ignored, as in the reference (loss.py:43-49)."""
    if sem_labels.dtype != torch.uint8:
        sem_labels = sem_labels.to(torch.uint8)
    loss = _StepBatchLoss.apply(alpha, color.contiguous(), pred_partfeat, gt_depth.contiguous(),
                                gt_color.contiguous(), sem_labels.contiguous(), z_vals.contiguous(),
                                float(color_scaling), float(opacity_scaling), float(partfeat_scaling), gt_partfeat)
    return loss, None
