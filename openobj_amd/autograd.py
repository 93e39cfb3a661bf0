"""torch.autograd.Functions over the backward entries of the C ABI (objnerf_mlp_backward_ws, objnerf_embed_bwd): a
caller that keeps the reference's loop body -- vmap(pe_model) -> vmap(fc_model) -> loss.step_batch_loss ->
loss.backward() (train.py:424-436) -- gets the reference's gradients on the stacked parameter tensors.  The fused
objnerf_train_step (training_strategy == "hip") remains the fast path; this is the compatible one.

The parameter tensors are arguments of the Functions only so that autograd routes gradients to them: the kernels read
the arena those tensors are views of."""
import torch

from . import ops


def _check_inputs(arena, params, what):
    for p in params:
        if torch.is_tensor(p) and not arena.owns(p):
            raise ValueError(f"{what}: a parameter tensor is not a view of the arena the kernels read")


def _check_unmodified(ctx, what):
    """backward() recomputes from the LIVE arena (only emb / pts are saved): an optimiser step, a copy-in or a scale
    change between forward and backward would give silently wrong gradients where torch raises on a modified saved
    tensor -- so does this."""
    if ctx.arena.state_version() != ctx.arena_version:
        raise RuntimeError(f"{what}: one of the variables needed for gradient computation has been modified by an "
                           "inplace operation (the parameter arena or its scale changed between forward and backward)")


class MlpFunction(torch.autograd.Function):
    """OccupancyMap.forward (model.py:61-103) of K stacked networks: emb [K,N,129] -> alpha [K,N], color [K,N,3],
    clip [K,N,C] (or an empty tensor).  `stacked`: params carry the leading K axis (vmap) or not (one module)."""

    @staticmethod
    def forward(ctx, arena, want_clip, stacked, emb, *params):
        _check_inputs(arena, params, "MlpFunction")
        alpha, color, _, clip = ops.mlp_forward(arena, emb, want_clip=want_clip)
        ctx.arena_version = arena.state_version()
        ctx.arena, ctx.want_clip, ctx.stacked, ctx.n_params = arena, want_clip, stacked, len(params)
        ctx.save_for_backward(emb)
        ctx.set_materialize_grads(False)
        return alpha, color, (clip if want_clip else emb.new_empty(0))

    @staticmethod
    def backward(ctx, d_alpha, d_color, d_clip):
        _check_unmodified(ctx, "MlpFunction")
        (emb,) = ctx.saved_tensors
        K, N = emb.shape[0], emb.shape[1]
        if d_alpha is None:
            d_alpha = emb.new_zeros(K, N)
        if d_color is None:
            d_color = emb.new_zeros(K, N, 3)
        if not ctx.want_clip:
            d_clip = None
        grads, d_emb = ops.mlp_backward(ctx.arena, emb, d_alpha, d_color, d_clip)
        views = ctx.arena.views(grads)
        out = []
        for i in range(ctx.n_params):
            if i in ops.FEAT_TENSORS and d_clip is None:
                out.append(None)                         # "no gradient", like an unused branch under autograd
            else:
                out.append(views[i] if ctx.stacked else views[i][0])
        return (None, None, None, d_emb if ctx.needs_input_grad[3] else None, *out)


class EmbedFunction(torch.autograd.Function):
    """UniDirsEmbed.forward (embedding.py:46-55) of K stacked embeddings: pts [K,N,3] -> [K,N,129]; differentiable
    w.r.t. B_layer.weight only (the reference never differentiates the sample positions)."""

    @staticmethod
    def forward(ctx, arena, stacked, pts, B):
        _check_inputs(arena, (B,), "EmbedFunction")
        ctx.arena, ctx.stacked = arena, stacked
        ctx.arena_version = arena.state_version()
        ctx.save_for_backward(pts)
        return ops.embed(arena, pts)

    @staticmethod
    def backward(ctx, d_emb):
        if ctx.needs_input_grad[2]:
            raise NotImplementedError("gradient w.r.t. the sample positions is not part of the training path")
        _check_unmodified(ctx, "EmbedFunction")
        (pts,) = ctx.saved_tensors
        d_B = ops.embed_backward(ctx.arena, pts, d_emb.contiguous())
        return None, None, None, (d_B if ctx.stacked else d_B[0])
