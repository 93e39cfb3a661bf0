"""utils -- the hot-path helpers of the reference's utils.py: update_vmap (:55-62), origin_dirs_W
(:324-336), ray_box_intersection (:309-319), stratified_bins (:342-379), normal_bins_sampling
(:382-397), performance_measure (:13-27).  Bbox / instance-tracking helpers are out of scope."""
from time import perf_counter_ns

import torch

from . import ops


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
        a = fmodel.arena
        K = a.K
        with torch.no_grad():
            if fmodel.kind == "pe":
                lead = x.shape[1:-1]
                return ops.embed(a, x.reshape(K, -1, 3).contiguous()).reshape(K, *lead, -1)
            lead = x.shape[1:-1]
            alpha, color, _, clip = ops.mlp_forward(a, x.reshape(K, -1, x.shape[-1]).contiguous(), want_clip=True)
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
    if optimiser is not None and hasattr(optimiser, "add_param_group"):
        optimiser.add_param_group({"params": params})
    return StackedModel(arena, "pe" if is_pe else "fc"), params, buffers


def ray_box_intersection(origins, directions, bounds_min, bounds_max):   # utils.py:309-319
    tmin = (bounds_min - origins) / directions
    tmax = (bounds_max - origins) / directions
    t1 = torch.min(tmin, tmax)
    t2 = torch.max(tmin, tmax)
    near = torch.amax(t1, dim=1)
    far = torch.amin(t2, dim=1)
    hit = (near <= far) & (far > 0)
    return near, far, hit


def origin_dirs_W(T_WC, dirs_C):                                          # utils.py:324-336
    assert T_WC.shape[0] == dirs_C.shape[0]
    assert T_WC.shape[1:] == (4, 4)
    if dirs_C.shape[1] == 3 and dirs_C.dim() == 2:
        dirs_W = torch.matmul(T_WC[:, :3, :3], dirs_C.unsqueeze(-1)).squeeze(-1)
    else:
        dirs_W = (T_WC[:, None, :3, :3] @ dirs_C[..., None]).squeeze()
    return T_WC[:, :3, -1], dirs_W


def stratified_bins(min_depth, max_depth, n_bins, n_rays, type=torch.float32, device="cuda:0"):   # :342-379
    lim = torch.linspace(0, 1, n_bins + 1, dtype=type, device=device)
    if not torch.is_tensor(min_depth):
        min_depth = torch.ones(n_rays, dtype=type, device=device) * min_depth
    if not torch.is_tensor(max_depth):
        max_depth = torch.ones(n_rays, dtype=type, device=device) * max_depth
    depth_range = max_depth - min_depth
    lower = (depth_range[..., None] * lim + min_depth[..., None])[:, :-1]
    assert lower.shape == (n_rays, n_bins)
    inc = torch.rand(n_rays, n_bins, device=device, dtype=torch.float32) * (depth_range / n_bins)[..., None]
    return lower + inc


def normal_bins_sampling(depth, n_bins, n_rays, delta, device="cuda:0"):                          # :382-397
    bins = torch.empty(n_rays, n_bins, dtype=torch.float32, device=device).normal_(mean=0., std=delta / 3.)
    bins = torch.clip(bins.sort().values, -delta, delta)
    z_vals = depth[:, None] + bins
    assert z_vals.shape == (n_rays, n_bins)
    return z_vals
