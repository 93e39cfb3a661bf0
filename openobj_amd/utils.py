"""utils -- the hot-path helpers of the reference's utils.py: update_vmap (:55-62), origin_dirs_W
(:324-336), ray_box_intersection (:309-319), stratified_bins (:342-379), normal_bins_sampling
(:382-397), performance_measure (:13-27).  Bbox / instance-tracking helpers are out of scope."""
from time import perf_counter_ns

import torch

from . import ops
from .autograd import EmbedFunction, MlpFunction


class performance_measure:
    """Wall-clock block timer; unlike the reference's it synchronises the device first, so GPU work is
    actually included."""

    def __init__(self, name) -> None:
        self.name = name

    def __enter__(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.start_time = perf_counter_ns()

    def __exit__(self, type, value, tb):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        self.end_time = perf_counter_ns()
        self.exec_time = self.end_time - self.start_time
        print(f"{self.name} excution time: {(self.exec_time)/1000000:.2f} ms")


class StackedModel:
    """What `combine_state_for_ensemble` returns as `fmodel`, for K arena-resident networks: call it
    through `vmap(fmodel)(params, buffers, x)` exactly like train.py:424-425."""

    def __init__(self, arena: ops.ParamArena, kind: str):
        self.arena, self.kind = arena, kind


def vmap(fmodel: StackedModel):
    """Stand-in for functorch.vmap on a StackedModel (train.py:424-425): one batched HIP launch."""
    def run(params, buffers, x):
        # differentiable w.r.t. `params` (and, for the networks, the embedding): loss.backward() of the reference's loop
        # body (train.py:424-436) reaches the stacked tensors through objnerf_mlp_backward_ws / objnerf_embed_bwd
        a = fmodel.arena
        K = a.K
        lead = x.shape[1:-1]
        if fmodel.kind == "pe":
            emb = EmbedFunction.apply(a, True, x.reshape(K, -1, 3).contiguous(), params[0])
            return emb.reshape(K, *lead, -1)
        alpha, color, clip = MlpFunction.apply(a, True, True, x.reshape(K, -1, x.shape[-1]).contiguous(), *params)
        return alpha.reshape(K, *lead, 1), color.reshape(K, *lead, 3), clip.reshape(K, *lead, -1)
    return run


def update_vmap(models, optimiser=None, arena=None):
    """Stack K per-object modules (all OccupancyMap, or all UniDirsEmbed) into ONE arena and return
    `(fmodel, params, buffers)` like utils.py:55-62.  params are the stacked [K,...] tensors (views of
    the arena).  Pass the arena returned for the FC models when stacking the PE models so both live in
    the same object-major block.  `optimiser`, when given, must offer add_param_group."""
    from .embedding import UniDirsEmbed
    K = len(models)
    is_pe = isinstance(models[0], UniDirsEmbed)
    src = models[0]._arena
    if arena is None:
        arena = ops.ParamArena(K, src.net, src.params.device)
    views = arena.views()
    with torch.no_grad():
        if is_pe:
            for k, m in enumerate(models):
                views[18][k].copy_(m.B_layer.weight)
                arena.scale[k] = float(m.scale)
            params = [views[18]]
            buffers = [torch.stack([m.frequency_bands for m in models]).to(arena.params.device), arena.scale]
        else:
            for k, m in enumerate(models):
                for i, p in enumerate(m.parameters()):
                    views[i][k].copy_(p)
            params = list(views[:18])
            buffers = []
    for p in params:                       # utils.py:58: [p.requires_grad_() for p in params]
        p.requires_grad_()
    if optimiser is not None and hasattr(optimiser, "add_param_group"):
        optimiser.add_param_group({"params": params})
    return StackedModel(arena, "pe" if is_pe else "fc"), params, buffers


def ray_box_intersection(origins, directions, bounds_min, bounds_max):
    """Slab test of [n,3] rays against an axis-aligned box -> (near, far, hit) (reference utils.py:309-319)."""
    return ops.ray_box(origins, directions, bounds_min, bounds_max)


def origin_dirs_W(T_WC, dirs_C):
    """Camera poses [F,4,4] and camera-frame ray directions [F,3] or [F,P,3] -> (origins [F,3], world directions)
    (reference utils.py:324-336); the rotation runs in objnerf_dirs_w, the origins are a view of the poses."""
    assert T_WC.shape[0] == dirs_C.shape[0]
    assert T_WC.shape[1:] == (4, 4)
    return T_WC[:, :3, -1], ops.dirs_w(T_WC, dirs_C)


def stratified_bins(min_depth, max_depth, n_bins, n_rays, type=torch.float32, device="cuda:0", u=None):
    """One uniform draw per bin of an even partition of [min_depth, max_depth] (scalars or per-ray tensors) ->
    [n_rays, n_bins] (reference utils.py:342-379).  `u`: injected uniforms; otherwise drawn by the counter-based
    generator of the library under torch's seed."""
    return ops.stratified_bins(min_depth, max_depth, int(n_bins), int(n_rays), device, u=u)


def normal_bins_sampling(depth, n_bins, n_rays, delta, device="cuda:0", g=None):
    """Sorted N(0, (delta/3)^2) offsets clipped to +-delta around each ray's depth -> [n_rays, n_bins] (reference
    utils.py:382-397).  `g`: injected draws."""
    assert depth.shape[0] == n_rays
    return ops.normal_bins(depth.to(device), int(n_bins), float(delta), g=g)
