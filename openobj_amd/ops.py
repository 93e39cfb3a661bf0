"""Thin tensor-level wrappers over the C ABI (include/objnerf_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; all arithmetic of the path runs in
libobjnerf_hip.so.  Every function requires CUDA(HIP) tensors and raises on anything else.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import AdamWArgs, LossArgs, Net, ObjnerfError, SampleArgs, TRAIN_SELF_COUNTS, TrainArgs, check, lib

EMB1, EMB2, N_DIRS = 87, 42, 21
TENSOR_NAMES = [
    "in_layer.0.weight", "in_layer.0.bias", "mid1.0.0.weight", "mid1.0.0.bias",
    "cat_layer.0.weight", "cat_layer.0.bias", "mid2.0.0.weight", "mid2.0.0.bias",
    "out_alpha.weight", "out_alpha.bias", "color_linear.0.weight", "color_linear.0.bias",
    "out_color.weight", "out_color.bias", "clip_linear.0.weight", "clip_linear.0.bias",
    "out_clip.weight", "out_clip.bias", "B_layer.weight",
]
FEAT_TENSORS = (14, 15, 16, 17)


def tensor_shapes(hidden: int, feat_dim: int = 512) -> List[Tuple[int, ...]]:
    H, Cc = hidden, feat_dim
    return [(H, EMB1), (H,), (H, H), (H,), (H, H + EMB1), (H,), (H, H), (H,), (1, H), (1,),
            (H, EMB2 + H), (H,), (3, H), (3,), (H, EMB2 + H), (H,), (Cc, H), (Cc,), (N_DIRS, 3)]


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    return t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _req(t: torch.Tensor, dtype, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.ObjnerfError(f"{name}: expected a GPU tensor (there is no CPU path)")
    if t.dtype != dtype:
        raise _lib.ObjnerfError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        t = t.contiguous()
    return t


@dataclass
class NetShape:
    hidden: int = 32
    feat_dim: int = 512
    n_freqs: int = 6

    def c(self) -> Net:
        return Net(self.hidden, self.feat_dim, self.n_freqs, 0)


class ParamArena:
    """K networks in one object-major fp32 arena [K, p_stride] (+ same-layout grad / Adam buffers).

    `views(buf)` returns the 19 stacked tensors the reference exposes after
    functorch.combine_state_for_ensemble (utils.py:55-62) as strided views of the arena.
    """

    def __init__(self, K: int, net: NetShape, device):
        self.K, self.net = K, net
        self.offsets, self.p_stride = _lib.param_layout(net.hidden, net.feat_dim, net.n_freqs)
        self.P = self.offsets[-1]
        self.shapes = tensor_shapes(net.hidden, net.feat_dim)
        self.params = torch.zeros(K, self.p_stride, dtype=torch.float32, device=device)
        self.scale = torch.full((K,), 2.0, dtype=torch.float32, device=device)
        self.version = 0            # bumped by every writer that goes around torch (the kernels write raw pointers)

    def set_scale(self, value: float) -> None:
        """scale[:] = value, skipped when that is what the last set_scale wrote and nobody has touched the tensor since
        (a module's forward calls this every time; an unconditional fill_ would count as a modification between the
        forward and the backward of an earlier call)."""
        key = (float(value), self.scale._version)
        if getattr(self, "_scale_set", None) == key:
            return
        self.scale.fill_(float(value))
        self._scale_set = (float(value), self.scale._version)

    def state_version(self):
        """What an autograd node compares between its forward and its backward (autograd.py): the kernels' own write
        counter plus torch's in-place counters of the two tensors the kernels read."""
        return (self.version, self.params._version, self.scale._version)

    def owns(self, t: torch.Tensor) -> bool:
        """t is a view into this arena's parameter block."""
        lo = self.params.data_ptr()
        return t.device == self.params.device and lo <= t.data_ptr() < lo + self.params.numel() * 4

    def views(self, buf: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        buf = self.params if buf is None else buf
        out = []
        for i, shp in enumerate(self.shapes):
            n = 1
            for s in shp:
                n *= s
            out.append(buf[:, self.offsets[i]:self.offsets[i] + n].view(self.K, *shp))
        return out

    def load_stacked(self, tensors: Sequence[torch.Tensor]) -> None:
        """tensors: 19 stacked [K,...] tensors (18 FC in parameters() order + B)."""
        for v, t in zip(self.views(), tensors):
            v.copy_(t.to(v.device))
        self.version += 1

    def has_grad_mask(self, with_feat: bool) -> torch.Tensor:
        m = torch.ones(self.P, dtype=torch.uint8, device=self.params.device)
        if not with_feat:
            m[self.offsets[14]:self.offsets[18]] = 0
        return m


# ---------------------------------------------------------------------------------------------------
def eval_points(arena: ParamArena, pts: torch.Tensor, want_hfeat: bool = False, want_clip: bool = False):
    """pts [K,N,3] -> alpha [K,N], color [K,N,3], hfeat [K,N,H] | None, clip [K,N,C] | None."""
    pts = _req(pts, torch.float32, "pts")
    K, N = pts.shape[0], pts.shape[1]
    dev = pts.device
    alpha = torch.empty(K, N, device=dev)
    color = torch.empty(K, N, 3, device=dev)
    hfeat = torch.empty(K, N, arena.net.hidden, device=dev) if (want_hfeat or want_clip) else None
    clip = torch.empty(K, N, arena.net.feat_dim, device=dev) if want_clip else None
    net = arena.net.c()
    nbytes = int(lib().objnerf_eval_workspace_bytes(C.byref(net), K, N))      # 0 for hidden 32 (fused kernel)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    check(lib().objnerf_eval_points_ws(C.byref(net), K, N, _ptr(arena.params), arena.p_stride, _ptr(arena.scale),
                                       _ptr(pts), _ptr(alpha), _ptr(color), _ptr(hfeat), _ptr(clip), _ptr(ws), nbytes,
                                       _stream()),
          "objnerf_eval_points_ws")
    return alpha, color, hfeat, clip


def box_rays(T_WC: torch.Tensor, T_OC: torch.Tensor, half_extent: torch.Tensor, dirs_C: torch.Tensor):
    """Camera rays of one view against an oriented box (trainer.py:136-167): dirs_C [P,3] ->
    dirs_W [P,3], near [P] (>= 0), far [P] (+0.2), hit [P] bool."""
    dirs_C = _req(dirs_C, torch.float32, "dirs_C")
    dev = dirs_C.device
    P = dirs_C.shape[0]
    T_WC = T_WC.to(dev, torch.float32).contiguous()
    T_OC = T_OC.to(dev, torch.float32).contiguous()
    half_extent = half_extent.to(dev, torch.float32).contiguous()
    dirs_W = torch.empty(P, 3, device=dev)
    near = torch.empty(P, device=dev)
    far = torch.empty(P, device=dev)
    hit = torch.empty(P, dtype=torch.uint8, device=dev)
    check(lib().objnerf_box_rays(P, _ptr(T_WC), _ptr(T_OC), _ptr(half_extent), _ptr(dirs_C), _ptr(dirs_W), _ptr(near),
                                 _ptr(far), _ptr(hit), _stream()), "objnerf_box_rays")
    return dirs_W, near, far, hit.bool()


def box_points(origin: torch.Tensor, dirs_W: torch.Tensor, near: torch.Tensor, far: torch.Tensor,
               u: Optional[torch.Tensor] = None, n_bins: Optional[int] = None, seed: Optional[int] = None,
               draw: Optional[int] = None):
    """Mid-points of the stratified bins of [near, far] (trainer.py:171-176): u [n, n_bins] (injected draws; None:
    generated in the kernel under `seed` and a per-call counter, n_bins required) -> z_vals [n, n_bins-1],
    pts [n, n_bins-1, 3]."""
    dirs_W = _req(dirs_W, torch.float32, "dirs_W")
    near = _req(near, torch.float32, "near")
    far = _req(far, torch.float32, "far")
    dev = dirs_W.device
    if u is not None:
        u = _req(u, torch.float32, "u")
        n, n_bins = u.shape
    else:
        n, n_bins = near.shape[0], int(n_bins)
    origin = origin.to(dev, torch.float32).contiguous()
    z = torch.empty(n, n_bins - 1, device=dev)
    pts = torch.empty(n, n_bins - 1, 3, device=dev)
    check(lib().objnerf_box_points(n, n_bins, _ptr(origin), _ptr(dirs_W), _ptr(near), _ptr(far), _ptr(u), _seed_of(seed),
                                   0 if u is not None else (_next_offset() if draw is None else int(draw)) & 0x1FFFFFFF,
                                   _ptr(z), _ptr(pts), _stream()),
          "objnerf_box_points")
    return z, pts


def render_fwd(arena: ParamArena, origin: torch.Tensor, dirs_W: torch.Tensor, near: torch.Tensor, far: torch.Tensor,
               u: Optional[torch.Tensor] = None, n_bins: Optional[int] = None, seed: Optional[int] = None,
               draw: Optional[int] = None, want_hfeat: bool = True, want_z: bool = False, bf16: bool = False):
    """The per-object chain of sceneObject.render_2D_syn (vmap.py:644-676) in ONE launch (objnerf_render_fwd, hidden 32):
    mid-points of the stratified bins of [near, far] -> embedding -> network -> compositing, a lane per ray, nothing per
    sample in HBM.  arena: ONE object (K = 1).  u [n, n_bins]: injected draws; None: the Philox draws of box_points under
    (seed, draw).  bf16: opt-in OBJNERF_TRAIN_BF16 arithmetic (bf16 MFMA operands, hardware sin / cos / sigmoid; not the
    reference's fp32).  -> dict(depth [n], opacity [n], rgb [n,3], vals [n,H] | None (composited feature hidden), z | None)."""
    if arena.K != 1 or arena.net.hidden != 32:
        raise ObjnerfError("render_fwd: one hidden-32 object per call")
    dirs_W = _req(dirs_W, torch.float32, "dirs_W")
    near, far = _req(near, torch.float32, "near"), _req(far, torch.float32, "far")
    dev = dirs_W.device
    if u is not None:
        u = _req(u, torch.float32, "u")
        n, n_bins = u.shape
    else:
        n, n_bins = near.shape[0], int(n_bins)
    origin = origin.to(dev, torch.float32).contiguous()
    depth, opacity = torch.empty(n, device=dev), torch.empty(n, device=dev)
    rgb = torch.empty(n, 3, device=dev)
    hf = torch.empty(n, arena.net.hidden, device=dev) if want_hfeat else None
    z = torch.empty(n, n_bins - 1, device=dev) if want_z else None
    net = arena.net.c()
    check(lib().objnerf_render_fwd(C.byref(net), n, n_bins, _ptr(arena.params), _ptr(arena.scale), _ptr(origin), _ptr(dirs_W),
                                   _ptr(near), _ptr(far), _ptr(u), _seed_of(seed),
                                   0 if u is not None else (_next_offset() if draw is None else int(draw)) & 0x1FFFFFFF,
                                   _ptr(depth), _ptr(opacity), _ptr(rgb), _ptr(hf), _ptr(z), _lib.TRAIN_BF16 if bf16 else 0,
                                   _stream()),
          "objnerf_render_fwd")
    return {"depth": depth, "opacity": opacity, "rgb": rgb, "vals": hf, "z": z}


def mlp_forward(arena: ParamArena, emb: torch.Tensor, want_hfeat: bool = False, want_clip: bool = False):
    """emb [K,N,129] -> alpha [K,N], color [K,N,3], hfeat | None, clip | None (OccupancyMap.forward)."""
    emb = _req(emb, torch.float32, "emb")
    K, N = emb.shape[0], emb.shape[1]
    dev = emb.device
    alpha = torch.empty(K, N, device=dev)
    color = torch.empty(K, N, 3, device=dev)
    hfeat = torch.empty(K, N, arena.net.hidden, device=dev) if (want_hfeat or want_clip) else None
    clip = torch.empty(K, N, arena.net.feat_dim, device=dev) if want_clip else None
    net = arena.net.c()
    nbytes = int(lib().objnerf_eval_workspace_bytes(C.byref(net), K, N))      # 0 for hidden 32 (fused kernel)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev) if nbytes else None
    check(lib().objnerf_mlp_forward_ws(C.byref(net), K, N, _ptr(arena.params), arena.p_stride, _ptr(emb), _ptr(alpha),
                                       _ptr(color), _ptr(hfeat), _ptr(clip), _ptr(ws), nbytes, _stream()),
          "objnerf_mlp_forward_ws")
    return alpha, color, hfeat, clip


def embed(arena: ParamArena, pts: torch.Tensor) -> torch.Tensor:
    pts = _req(pts, torch.float32, "pts")
    K, N = pts.shape[0], pts.shape[1]
    E = 3 + N_DIRS * arena.net.n_freqs
    out = torch.empty(K, N, E, device=pts.device)
    net = arena.net.c()
    check(lib().objnerf_embed(C.byref(net), K, N, _ptr(arena.params), arena.p_stride, _ptr(arena.scale),
                              _ptr(pts), _ptr(out), _stream()), "objnerf_embed")
    return out


def mlp_backward(arena: ParamArena, emb: torch.Tensor, d_alpha: torch.Tensor, d_color: torch.Tensor,
                 d_clip: Optional[torch.Tensor] = None):
    """Backward of OccupancyMap.forward (model.py:61-103) for the K stacked networks: -> (grads [K,p_stride] in arena
    layout -- tensors 0..13, 0..17 with d_clip, the rest zero --, d_emb [K,N,129])."""
    emb = _req(emb, torch.float32, "emb")
    K, N = emb.shape[0], emb.shape[1]
    d_alpha = _req(d_alpha.reshape(K, N), torch.float32, "d_alpha")
    d_color = _req(d_color.reshape(K, N, 3), torch.float32, "d_color")
    if d_clip is not None:
        d_clip = _req(d_clip.reshape(K, N, arena.net.feat_dim), torch.float32, "d_clip")
    dev = emb.device
    grads = torch.zeros(K, arena.p_stride, device=dev)
    d_emb = torch.empty_like(emb)
    net = arena.net.c()
    nbytes = int(lib().objnerf_mlp_backward_workspace_bytes(C.byref(net), K, N, 1 if d_clip is not None else 0))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    check(lib().objnerf_mlp_backward_ws(C.byref(net), K, N, _ptr(arena.params), arena.p_stride, _ptr(emb), _ptr(d_alpha),
                                        _ptr(d_color), _ptr(d_clip), _ptr(grads), _ptr(d_emb), _ptr(ws), nbytes, _stream()),
          "objnerf_mlp_backward_ws")
    return grads, d_emb


def embed_backward(arena: ParamArena, pts: torch.Tensor, d_emb: torch.Tensor) -> torch.Tensor:
    """Backward of UniDirsEmbed.forward (embedding.py:46-55) w.r.t. B_layer.weight: -> d B [K,21,3]."""
    pts = _req(pts, torch.float32, "pts")
    K, N = pts.shape[0], pts.shape[1]
    d_emb = _req(d_emb.reshape(K, N, -1), torch.float32, "d_emb")
    d_B = torch.empty(K, N_DIRS, 3, device=pts.device)
    scratch = torch.empty(K * 64, device=pts.device)
    net = arena.net.c()
    check(lib().objnerf_embed_bwd(C.byref(net), K, N, _ptr(arena.params), arena.p_stride, _ptr(arena.scale), _ptr(pts),
                                  _ptr(d_emb), _ptr(d_B), _ptr(scratch), _stream()), "objnerf_embed_bwd")
    return d_B


def occupancy(alpha: torch.Tensor) -> torch.Tensor:
    alpha = _req(alpha, torch.float32, "alpha")
    out = torch.empty_like(alpha)
    check(lib().objnerf_occupancy(alpha.numel(), _ptr(alpha), _ptr(out), _stream()), "objnerf_occupancy")
    return out


def composite(alpha: torch.Tensor, color: Optional[torch.Tensor], z: Optional[torch.Tensor],
              vals: Optional[torch.Tensor] = None, want_term: bool = False,
              input_is_occupancy: bool = False) -> Dict[str, torch.Tensor]:
    """alpha [n,S], color [n,S,3], z [n,S], vals [n,S,V] -> term/depth/var/rgb/opacity/vals."""
    alpha = _req(alpha, torch.float32, "alpha")
    if z is not None:
        z = _req(z, torch.float32, "z")
    n, S = alpha.shape
    dev = alpha.device
    if color is not None:
        color = _req(color, torch.float32, "color")
    V = 0
    out_vals = None
    if vals is not None:
        vals = _req(vals, torch.float32, "vals")
        V = vals.shape[-1]
        out_vals = torch.empty(n, V, device=dev)
    term = torch.empty(n, S, device=dev) if want_term else None
    depth = torch.empty(n, device=dev) if z is not None else None
    var = torch.empty(n, device=dev) if z is not None else None
    rgb = torch.empty(n, 3, device=dev) if color is not None else None
    opacity = torch.empty(n, device=dev)
    check(lib().objnerf_composite(n, S, int(input_is_occupancy), _ptr(alpha), _ptr(color), _ptr(z), _ptr(vals), V, _ptr(term), _ptr(depth),
                                  _ptr(var), _ptr(rgb), _ptr(opacity), _ptr(out_vals), _stream()),
          "objnerf_composite")
    return dict(term=term, depth=depth, var=var, rgb=rgb, opacity=opacity, vals=out_vals)


def feature_head(arena: ParamArena, hfeat: torch.Tensor, weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    """hfeat [K,n,H] (+ weight [K,n]) -> [K,n,C] = out_clip(.) applied after compositing."""
    hfeat = _req(hfeat, torch.float32, "hfeat")
    K, n = hfeat.shape[0], hfeat.shape[1]
    if weight is not None:
        weight = _req(weight, torch.float32, "weight")
    out = torch.empty(K, n, arena.net.feat_dim, device=hfeat.device)
    net = arena.net.c()
    check(lib().objnerf_feature_head(C.byref(net), K, n, _ptr(arena.params), arena.p_stride, _ptr(hfeat),
                                     _ptr(weight), _ptr(out), _stream()), "objnerf_feature_head")
    return out


def label_counts(labels: torch.Tensor):
    labels = _req(labels, torch.uint8, "labels")
    K, R = labels.shape
    counts = torch.empty(K, 2, dtype=torch.int32, device=labels.device)
    flags = torch.empty(2, dtype=torch.int32, device=labels.device)
    check(lib().objnerf_label_counts(K, R, _ptr(labels), _ptr(counts), _ptr(flags), _stream()),
          "objnerf_label_counts")
    return counts, flags


def step_batch_loss(alpha, color, gt_depth, gt_rgb, labels, z, color_scaling=5.0, opacity_scaling=10.0,
                    gt_feat=None, pred_feat=None, feat_scaling=5.0, want_grads=True, flags_in=None, counts_in=None):
    """loss.step_batch_loss on materialised tensors.  Returns dict(total, terms [K,4], d_alpha, d_color,
    d_pred_feat, status)."""
    alpha = _req(alpha, torch.float32, "alpha")
    K, R, S = alpha.shape[0], alpha.shape[1], alpha.shape[2]
    color = _req(color, torch.float32, "color")
    z = _req(z, torch.float32, "z")
    gt_depth = _req(gt_depth, torch.float32, "gt_depth")
    gt_rgb = _req(gt_rgb, torch.float32, "gt_rgb")
    labels = _req(labels, torch.uint8, "labels")
    dev = alpha.device
    Cc = 0
    if gt_feat is not None:
        gt_feat = _req(gt_feat, torch.float32, "gt_feat")
        pred_feat = _req(pred_feat, torch.float32, "pred_feat")
        Cc = gt_feat.shape[-1]
    terms = torch.empty(K, 4, device=dev)
    total = torch.empty(1, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    counts = torch.empty(2 * K + 2, dtype=torch.int32, device=dev)
    d_alpha = torch.empty(K, R, S, device=dev) if want_grads else None
    d_color = torch.empty(K, R, S, 3, device=dev) if want_grads else None
    d_pf = torch.empty_like(pred_feat) if (want_grads and pred_feat is not None) else None
    a = LossArgs(K, R, S, Cc, color_scaling, opacity_scaling, feat_scaling, 0.0, _ptr(alpha), _ptr(color), _ptr(z),
                 _ptr(gt_depth), _ptr(gt_rgb), _ptr(labels), _ptr(pred_feat), _ptr(gt_feat), _ptr(flags_in), _ptr(counts_in),
                 _ptr(terms), _ptr(total), _ptr(d_alpha), _ptr(d_color), _ptr(d_pf), _ptr(counts), _ptr(status))
    check(lib().objnerf_step_batch_loss(C.byref(a), _stream()), "objnerf_step_batch_loss")
    return dict(total=total, terms=terms, d_alpha=d_alpha, d_color=d_color, d_pred_feat=d_pf, status=status,
                counts=counts)


# bytes; the layer-wise / hidden-256 paths materialise activations per object chunk (OBJNERF_WORKSPACE_BUDGET_GIB: override)
LAYERWISE_WORKSPACE_BUDGET = int(float(os.environ.get("OBJNERF_WORKSPACE_BUDGET_GIB", "64")) * (1 << 30))


def precision_bits(p) -> int:
    """The opt-in operand precision of a training step -> objnerf_train_args.mode bits.  False / None / "fp32": the
    reference's arithmetic; True / "bf16": OBJNERF_TRAIN_BF16; "fp16": OBJNERF_TRAIN_FP16 (layer-wise path)."""
    if p in (False, None, "fp32"):
        return 0
    if p in (True, "bf16"):
        return 1
    if p == "fp16":
        return 4
    raise ValueError("precision must be fp32 / bf16 / fp16, not {!r}".format(p))


class StreamContext:
    """objnerf_context: the helper streams + events the layer-wise path forks onto.  Created on the current device;
    freed with the object (after a device synchronisation, so no step using it is in flight)."""

    def __init__(self, device):
        self.device = torch.device(device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            check(lib().objnerf_context_create(C.byref(h)), "objnerf_context_create")
        self.handle = h

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        if h:
            try:
                torch.cuda.synchronize(self.device)
                lib().objnerf_context_destroy(h)
            except Exception:      # interpreter shutdown
                pass


class TrainWorkspace:
    """Caller-owned buffers of the fused training step, allocated once per (K,R,S).

    The layer-wise path (hidden != 32, S > 64 or layerwise=True) keeps every activation of the objects it works on
    in the workspace; when K objects would need more than `budget` bytes (BASELINE configs[4]: 64 objects x 8192
    rays x 128 samples x hidden 256 is ~0.7 TB), the workspace is sized for `k_chunk` objects and train_step runs
    the objects chunk by chunk -- they are independent networks, only the early-return flags span the batch."""

    def __init__(self, arena: ParamArena, K: int, R: int, S: int, with_feat: bool, layerwise: bool = False,
                 budget: Optional[int] = None, precision=None, context: Optional["StreamContext"] = None):
        """context: a StreamContext to share (the K one-object workspaces of the forloop strategy run one after the other
        on one stream, so one set of helper streams serves them all); None = this workspace owns one when its path uses it."""
        layerwise = layerwise or precision_bits(precision) == 4          # the fp16 mode lives on the layer-wise path
        dev = arena.params.device
        net = arena.net.c()
        wf = int(with_feat) | (2 if layerwise else 0) | (4 if precision_bits(precision) in (1, 4) else 0)   # bit 2: 16-bit modes only
        budget = LAYERWISE_WORKSPACE_BUDGET if budget is None else budget
        self.k_chunk = K
        nbytes = lib().objnerf_train_workspace_bytes(C.byref(net), K, R, S, wf)
        if (arena.net.hidden != 32 or S > 64 or layerwise) and nbytes > budget and K > 1:
            per_obj = lib().objnerf_train_workspace_bytes(C.byref(net), 1, R, S, wf)
            self.k_chunk = max(1, min(K, int(budget // max(1, per_obj))))
            # (measured, round 6: EQUAL chunks -- 8 x 8 objects instead of 7 x 9 + 1 for configs[4]'s 64 -- are slower, 191.6
            # against 187.4 ms per step: the fused hidden-256 kernels gain more from a ninth object per launch than the
            # one-object tail costs)
            nbytes = lib().objnerf_train_workspace_bytes(C.byref(net), self.k_chunk, R, S, wf)
        if nbytes == 0:
            raise _lib.ObjnerfError("objnerf_train_workspace_bytes returned 0")
        self.nbytes = int(nbytes)
        # the layer-wise path runs independent GEMMs side by side on the context's streams
        uses_context = (arena.net.hidden != 32 or S > 64 or layerwise) and dev.type == "cuda"
        self.context = (context if context is not None else StreamContext(dev)) if uses_context else None
        self.buf = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        # Chunked hidden-256 steps in a 16-bit mode (configs[4]: the two fused kernels of objnerf_train256.hip) run their
        # chunks ALTERNATELY on two streams, each with its own buffer: a chunk's HBM-bound weight-gradient kernel then fills the
        # CUs that the next chunk's kernel A leaves idle at its tail (both hold ~150 KB of LDS per workgroup, so they never
        # share a CU; tools/c5_overlap.py: 3 - 5 % of the step).  Only where a second buffer fits beside the first.
        self.lanes = 1
        self.buf2 = None
        if (self.k_chunk < K and dev.type == "cuda" and arena.net.hidden == 256 and precision_bits(precision) in (1, 4)
                and not (layerwise and precision_bits(precision) != 4) and S in (32, 64, 128)
                and os.environ.get("OBJNERF_ONE_LANE", "0") != "1"):
            # k_chunk was sized for ONE buffer of up to `budget` bytes: the second one is taken only where it leaves a quarter
            # of the device (and at least 16 GiB) free for what comes after this workspace -- sample pools, render buffers,
            # keyframe stores -- instead of "8 GiB above the buffer" (round 5: ~128 GiB of workspace with 8 GiB of slack)
            free, total = torch.cuda.mem_get_info(dev)
            if free - self.nbytes > max(total // 4, 16 << 30):
                self.buf2 = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
                # TWO side streams created back to back (the caller's stream only hands over and collects): streams are dealt
                # to a few hardware queues round-robin in creation order, and two streams that share a queue run their kernels
                # back to back -- a single side stream overlapped with the caller's stream in a fresh process and not in one
                # that had created a dozen streams before; a high-priority side stream the other way round (tools/lanes_ab.sh)
                with torch.cuda.device(dev):        # (events belong to the device current at their creation)
                    self.sides = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
                    self.ev_in, self.ev_out = torch.cuda.Event(), [torch.cuda.Event(), torch.cuda.Event()]
                self.lanes = 2
        self.grads = torch.zeros_like(arena.params)
        self.loss_terms = torch.zeros(K, 4, device=dev)
        self.status = torch.zeros(1, dtype=torch.int32, device=dev)
        self.status_chunks = torch.zeros((K + self.k_chunk - 1) // self.k_chunk, dtype=torch.int32, device=dev)
        self.counts = torch.zeros(K, 2, dtype=torch.int32, device=dev)
        self.flags = torch.zeros(2, dtype=torch.int32, device=dev)
        # (the workspace size depends on the operand precision: a loop whose precision is toggled must re-allocate)
        self.key = self.make_key(K, R, S, with_feat, precision)

    @staticmethod
    def make_key(K, R, S, with_feat, precision=None):
        return (K, R, S, bool(with_feat), precision_bits(precision))


def train_step(arena: ParamArena, ws: TrainWorkspace, batch: Dict[str, torch.Tensor], color_scaling=5.0,
               opacity_scaling=10.0, feat_scaling=5.0, with_feat=False, obj_center=0.0,
               global_flags: Optional[torch.Tensor] = None, global_counts: Optional[torch.Tensor] = None,
               bf16: bool = False, layerwise: bool = False, relu_masks: Optional[torch.Tensor] = None,
               emb_debug: Optional[torch.Tensor] = None, optim=None) -> None:
    """One fused iteration (train.py:424-472): fills ws.grads, ws.loss_terms, ws.status.

    optim: an optim.ArenaAdamW over `arena` -- the iteration's optimiser.step() (train.py:472-473) runs INSIDE the call
    (objnerf_train_args.optim: the launch that reduces the partial gradients applies AdamW to each element it has just
    summed, with the iteration's early-return flags deciding which tensor groups are stepped); ws.grads is still
    written.  None: gradients only.

    Without global_flags AND global_counts the step derives the label statistics itself (OBJNERF_TRAIN_SELF_COUNTS;
    ws.counts / ws.flags receive them) -- no separate objnerf_label_counts call.

    layerwise: OBJNERF_TRAIN_LAYERWISE -- run the layer-wise (any width) implementation even where the fused
    kernel applies (cross-check of two independent implementations; ws must be built with layerwise=True).

    bf16: operand precision (precision_bits): True / "bf16" = OBJNERF_TRAIN_BF16 (bf16 MFMA operands, fp32 accumulate /
    master weights), "fp16" = OBJNERF_TRAIN_FP16 (layer-wise path; ws built with precision="fp16"); the default is
    the reference's fp32 arithmetic.

    relu_masks: test hook -- uint8 [K,R,S,6,hidden/8] receiving the ReLU branch bits of the iteration
    (objnerf_train_args.relu_masks); None in production.
    emb_debug: test hook -- float32 [K,R,S,129] receiving the embedding rows the fused fp32 kernel formed in registers
    (objnerf_train_args.emb_debug; hidden 32, S <= 64, fp32 only); None in production.

    batch: pts [K,R,S,3] (or origins+dirs), z, gt_depth, gt_rgb, labels u8 (+ gt_feat when with_feat).
    global_flags: optional [2] int32 tensor already max-reduced over all ranks (object sharding)."""
    z = _req(batch["z"], torch.float32, "z")
    K, R, S = z.shape
    labels = _req(batch["labels"], torch.uint8, "labels")
    pts = batch.get("pts")
    pts = _req(pts, torch.float32, "pts") if pts is not None else None
    origins = _req(batch["origins"], torch.float32, "origins") if pts is None else None
    dirs = _req(batch["dirs"], torch.float32, "dirs") if pts is None else None
    gt_depth = _req(batch["gt_depth"], torch.float32, "gt_depth")
    gt_rgb = _req(batch["gt_rgb"], torch.float32, "gt_rgb")
    gt_feat = _req(batch["gt_feat"], torch.float32, "gt_feat") if with_feat else None
    st = _stream()
    self_counts = global_flags is None and global_counts is None
    if not self_counts and (global_flags is None or global_counts is None):
        check(lib().objnerf_label_counts(K, R, _ptr(labels), _ptr(ws.counts), _ptr(ws.flags), st),
              "objnerf_label_counts")
    flags = ws.flags
    if global_flags is not None:
        flags = _req(global_flags, torch.int32, "global_flags")
    counts = ws.counts
    if global_counts is not None:       # one object's rays split over ranks (background): global mask counts
        counts = _req(global_counts, torch.int32, "global_counts")
    net = arena.net.c()
    mode = precision_bits(bf16) | (2 if layerwise else 0)
    if relu_masks is not None:
        relu_masks = _req(relu_masks, torch.uint8, "relu_masks")
        if tuple(relu_masks.shape) != (K, R, S, 6, arena.net.hidden // 8):
            raise ObjnerfError("relu_masks must be uint8 [K,R,S,6,hidden/8]")
    if emb_debug is not None:
        emb_debug = _req(emb_debug, torch.float32, "emb_debug")
        if tuple(emb_debug.shape) != (K, R, S, EMB1 + EMB2):
            raise ObjnerfError("emb_debug must be float32 [K,R,S,129]")
    kc = getattr(ws, "k_chunk", K)
    ctx = ws.context.handle if getattr(ws, "context", None) is not None else None
    if kc >= K:
        oa = None
        if optim is not None:
            if optim.arena is not arena:
                raise ObjnerfError("train_step: optim must be the optimiser of `arena`")
            oa = AdamWArgs(_ptr(optim.exp_avg), _ptr(optim.exp_avg_sq), _ptr(optim._banks), int(optim._bank), 0,
                           optim.lr, optim.betas[0], optim.betas[1], optim.eps, optim.weight_decay, 0.0)
        a = TrainArgs(K, R, S, mode | (TRAIN_SELF_COUNTS if self_counts else 0), color_scaling, opacity_scaling,
                      feat_scaling, obj_center, _ptr(arena.params),
                      arena.p_stride, _ptr(arena.scale), _ptr(pts), _ptr(origins), _ptr(dirs), _ptr(z), _ptr(gt_depth),
                      _ptr(gt_rgb), _ptr(labels), _ptr(gt_feat), _ptr(counts), _ptr(flags), _ptr(ws.grads),
                      _ptr(ws.loss_terms), _ptr(ws.status), _ptr(ws.buf), ws.nbytes, _ptr(relu_masks), ctx, _ptr(emb_debug),
                      C.addressof(oa) if oa is not None else None)
        check(lib().objnerf_train_step(C.byref(net), C.byref(a), st), "objnerf_train_step")
        if optim is not None:
            optim._bank ^= 1
            arena.version += 1          # (the call wrote the arena through raw pointers: torch's counter misses it)
        return
    if self_counts:                     # (the flags span the whole batch, not a chunk of it)
        check(lib().objnerf_label_counts(K, R, _ptr(labels), _ptr(ws.counts), _ptr(ws.flags), st),
              "objnerf_label_counts")
    # layer-wise path, chunk of objects at a time (leading-dimension slices are contiguous views)
    sl = lambda t, k0, k1: None if t is None else t[k0:k1]           # noqa: E731
    two = getattr(ws, "lanes", 1) == 2 and relu_masks is None and emb_debug is None
    if two:                             # (even / odd chunks on the two side streams, each with its buffer, behind this stream)
        main = torch.cuda.current_stream(z.device)
        ws.ev_in.record(main)
        for sd in ws.sides:
            sd.wait_event(ws.ev_in)
    for ci, k0 in enumerate(range(0, K, kc)):
        k1 = min(K, k0 + kc)
        buf = ws.buf2 if (two and (ci & 1)) else ws.buf
        sth = ws.sides[ci & 1].cuda_stream if two else st
        a = TrainArgs(k1 - k0, R, S, mode, color_scaling, opacity_scaling, feat_scaling, obj_center,
                      _ptr(arena.params[k0:k1]), arena.p_stride, _ptr(arena.scale[k0:k1]), _ptr(sl(pts, k0, k1)),
                      _ptr(sl(origins, k0, k1)), _ptr(sl(dirs, k0, k1)), _ptr(z[k0:k1]), _ptr(gt_depth[k0:k1]),
                      _ptr(gt_rgb[k0:k1]), _ptr(labels[k0:k1]), _ptr(sl(gt_feat, k0, k1)), _ptr(counts[k0:k1]),
                      _ptr(flags), _ptr(ws.grads[k0:k1]), _ptr(ws.loss_terms[k0:k1]), _ptr(ws.status_chunks[ci:ci + 1]),
                      _ptr(buf), ws.nbytes, _ptr(sl(relu_masks, k0, k1)), ctx, _ptr(sl(emb_debug, k0, k1)), None)
        check(lib().objnerf_train_step(C.byref(net), C.byref(a), sth), "objnerf_train_step")
    if two:
        for sd, ev in zip(ws.sides, ws.ev_out):
            ev.record(sd)
            main.wait_event(ev)
    torch.amax(ws.status_chunks, dim=0, keepdim=True, out=ws.status)
    if optim is not None:               # (chunked: one optimiser launch over the whole arena after the last chunk)
        optim.step(ws.grads, arena.has_grad_mask(with_feat), flags=flags)


def adamw_step(arena: ParamArena, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
               has_grad: Optional[torch.Tensor], step: int, lr: float, weight_decay: float, beta1=0.9, beta2=0.999,
               eps=1e-8, params: Optional[torch.Tensor] = None) -> None:
    p = arena.params if params is None else params
    if params is None:
        arena.version += 1              # (raw-pointer write: torch's version counter misses it; autograd.py checks this one)
    check(lib().objnerf_adamw_step(arena.K, arena.P, arena.p_stride, _ptr(p), _ptr(grads), _ptr(exp_avg),
                                   _ptr(exp_avg_sq), _ptr(has_grad), step, lr, beta1, beta2, eps, weight_decay,
                                   _stream()), "objnerf_adamw_step")


def adamw_step_flags(arena: ParamArena, grads: torch.Tensor, exp_avg: torch.Tensor, exp_avg_sq: torch.Tensor,
                     has_grad: Optional[torch.Tensor], flags: torch.Tensor, group_steps: torch.Tensor, bank: int,
                     lr: float, weight_decay: float, beta1=0.9, beta2=0.999, eps=1e-8,
                     params: Optional[torch.Tensor] = None) -> None:
    """AdamW with the iteration's early-return flags deciding on the device which tensor groups received a gradient
    (objnerf_adamw_step_flags); group_steps: int32[2, 3] device counters owned by the caller, `bank` the bank to read
    (the other one receives the advanced counters: alternate it from call to call)."""
    p = arena.params if params is None else params
    if params is None:
        arena.version += 1
    flags = _req(flags, torch.int32, "flags")
    group_steps = _req(group_steps, torch.int32, "group_steps")
    if group_steps.numel() != 6 or bank not in (0, 1):
        raise ObjnerfError("adamw_step_flags: group_steps must be int32[2, 3], bank 0 or 1")
    o = arena.offsets
    check(lib().objnerf_adamw_step_flags(arena.K, arena.P, arena.p_stride, _ptr(p), _ptr(grads), _ptr(exp_avg),
                                         _ptr(exp_avg_sq), _ptr(has_grad), _ptr(flags), _ptr(group_steps), int(bank),
                                         int(o[10]), int(o[14]), int(o[18]), lr, beta1, beta2, eps, weight_decay, _stream()),
          "objnerf_adamw_step_flags")


def ingest_frame(rgb, depth, inst, t_wc, items) -> None:
    """One frame into a keyframe slot of every visible object, one launch (train.py:196-256).
    rgb u8 [W,H,3], depth f32 [W,H], inst int32 [W,H], t_wc f32 [4,4] on the device;
    items = [(store tensors (rgbs_batch, depth_batch, t_wc_batch, bbox), slot, obj_id, box[4])]."""
    import numpy as np
    from ._lib import IngestItem
    rgb = _req(rgb, torch.uint8, "rgb")
    depth = _req(depth, torch.float32, "depth")
    inst = _req(inst, torch.int32, "inst")
    t_wc = _req(t_wc, torch.float32, "t_wc")
    W, H = depth.shape
    arr = (IngestItem * len(items))()
    for i, ((rgbs_b, depth_b, twc_b, bbox_b), slot, obj_id, box) in enumerate(items):
        if rgbs_b.shape[1:] != (W, H, 4) or depth_b.shape[1:] != (W, H) or not 0 <= slot < rgbs_b.shape[0]:
            raise ObjnerfError("ingest_frame: store / frame shapes do not match")
        arr[i] = IngestItem(rgbs_b.data_ptr(), depth_b.data_ptr(), twc_b.data_ptr(), bbox_b.data_ptr(), int(slot),
                            int(obj_id), (C.c_float * 4)(*[float(v) for v in box]))
    host = torch.from_numpy(np.frombuffer(arr, dtype=np.uint8).copy())
    dev_items = host.to(rgb.device)
    check(lib().objnerf_ingest_frame(W, H, _ptr(rgb), _ptr(depth), _ptr(inst), _ptr(t_wc), len(items), _ptr(dev_items),
                                     _stream()), "objnerf_ingest_frame")


def keyframe_table(stores) -> torch.Tensor:
    """Device descriptor table of objnerf_sample_rays_stacked: stores = [(rgbs_batch, depth_batch, t_wc_batch, bbox)]
    per object (the tensors must stay alive and in place while the table is used).  int64 [K, 4] of device pointers."""
    rows = []
    for rgbs, depth, t_wc, bbox in stores:
        _req(rgbs, torch.uint8, "rgbs_batch"); _req(depth, torch.float32, "depth_batch")
        _req(t_wc, torch.float32, "t_wc_batch"); _req(bbox, torch.float32, "bbox")
        rows.append([rgbs.data_ptr(), depth.data_ptr(), t_wc.data_ptr(), bbox.data_ptr()])
    return torch.tensor(rows, dtype=torch.int64).to(stores[0][0].device)


def _sample_common(K, n_frames, n_px, n_cam2surf, n_bins, dev, want_pts, record, stacked):
    n, S = n_frames * n_px, n_cam2surf + n_bins
    lead = (K,) if stacked else ()
    o = {"rgb": torch.empty(*lead, n, 3, dtype=torch.uint8, device=dev), "depth": torch.empty(*lead, n, device=dev),
         "valid": torch.empty(*lead, n, dtype=torch.uint8, device=dev),
         "labels": torch.empty(*lead, n, dtype=torch.uint8, device=dev), "z": torch.empty(*lead, n, S, device=dev),
         "pts": torch.empty(*lead, n, S, 3, device=dev) if want_pts else None,
         "origins": None if want_pts else torch.empty(*lead, n, 3, device=dev),
         "dirs": None if want_pts else torch.empty(*lead, n, 3, device=dev),
         "kf": torch.empty(*lead, n_frames, dtype=torch.int64, device=dev) if record else None,
         "px": torch.empty(*lead, n, 2, dtype=torch.int32, device=dev) if record else None,
         "ws": torch.empty(*lead, 1 + 6 * n, device=dev)}
    return o


def _partfeat_fields(a: SampleArgs, partfeat, K, F, n, dev, stacked):
    """ABI 6 fields of the part-feature gather (vmap.py:437-452).  partfeat = (global_partfeat [Fn, Wp, Hp, C] fp32,
    use_frame [F] | [K, F] (dataset frame id of every keyframe slot), stride, part_down) or None.  Returns the output
    tensor ([n, C] | [K, n, C]) the launch fills; the frame range is checked here (the reference would raise an
    IndexError from its advanced indexing, the kernel clamps)."""
    if partfeat is None:
        return None
    gpf, use_frame, stride, part_down = partfeat
    gpf = _req(gpf, torch.float32, "global_partfeat")
    if gpf.dim() != 4:
        raise ObjnerfError("global_partfeat must be [frames, W / part_down, H / part_down, C]")
    uf = torch.as_tensor(np.asarray(use_frame.cpu() if torch.is_tensor(use_frame) else use_frame))   # host bookkeeping
    if int(stride) != stride or int(stride) <= 0 or bool((uf != uf.round()).any()):
        raise ObjnerfError("part features: integer frame ids and stride expected")
    if uf.numel() != K * F:
        raise ObjnerfError("use_frame must hold one frame id per keyframe slot")
    if int(uf.min()) < 0 or int(uf.max()) // int(stride) >= gpf.shape[0]:
        raise IndexError("use_frame / stride outside global_partfeat")
    uf = uf.to(torch.int32).contiguous().to(dev)
    out = torch.empty(*((K,) if stacked else ()), n, gpf.shape[3], device=dev)
    a.global_partfeat, a.use_frame, a.out_partfeat = _ptr(gpf), _ptr(uf), _ptr(out)
    a.pf_frames, a.pf_w, a.pf_h, a.pf_c = gpf.shape
    a.pf_stride, a.part_down = int(stride), float(part_down)
    a._keep = (gpf, uf)
    return out


def _seed_of(seed):
    return (torch.initial_seed() if seed is None else int(seed)) & (2 ** 64 - 1)


def sample_rays_stacked(table: torch.Tensor, F: int, W: int, H: int, rays_dir_cache, kf_ids, u_w, u_h, u, g,
                        n_cam2surf: int, n_bins: int, surface_eps: float, stop_eps: float, min_bound: float = 0.0,
                        obj_center: float = 0.0, partfeat=None):
    """sample_rays for K objects in one launch chain, INJECTED draws: kf_ids [K, n_frames], u_w / u_h
    [K, n_frames, n_px], u [K, n, N+M], g [K, n, M].  Returns the STACKED batch tensors (rgb u8 [K,n,3], depth [K,n],
    valid [K,n] bool, labels u8 [K,n], pts [K,n,S,3], z [K,n,S])."""
    table = _req(table, torch.int64, "table")
    rays_dir_cache = _req(rays_dir_cache, torch.float32, "rays_dir_cache")
    kf_ids = _req(kf_ids, torch.int64, "kf_ids")
    u_w, u_h = _req(u_w, torch.float32, "u_w"), _req(u_h, torch.float32, "u_h")
    u, g = _req(u, torch.float32, "u"), _req(g, torch.float32, "g")
    K, n_frames, n_px = u_w.shape
    if table.shape != (K, 4) or kf_ids.shape != (K, n_frames):
        raise ObjnerfError("sample_rays_stacked: table / kf_ids do not match the draws")
    n = n_frames * n_px
    S = n_cam2surf + n_bins
    if u.shape != (K, n, S) or g.shape != (K, n, n_bins):
        raise ObjnerfError("sample_rays_stacked: u / g shapes")
    o = _sample_common(K, n_frames, n_px, n_cam2surf, n_bins, table.device, True, False, True)
    a = SampleArgs(F, W, H, n_frames, n_px, n_cam2surf, n_bins, 0, surface_eps, stop_eps, min_bound, obj_center,
                   None, None, None, None, _ptr(rays_dir_cache), _ptr(kf_ids), _ptr(u_w), _ptr(u_h), _ptr(u), _ptr(g),
                   _ptr(o["rgb"]), _ptr(o["depth"]), _ptr(o["valid"]), _ptr(o["labels"]), _ptr(o["z"]), _ptr(o["pts"]),
                   _ptr(o["ws"]), 0, 0, 0, None, None, None, None, None)
    pf = _partfeat_fields(a, partfeat, K, F, n, table.device, True)
    check(lib().objnerf_sample_rays_stacked(C.byref(a), K, _ptr(table), _stream()), "objnerf_sample_rays_stacked")
    r = (o["rgb"], o["depth"], o["valid"].bool(), o["labels"], o["pts"], o["z"])
    return r if partfeat is None else r + (pf,)


def sample_rays_seeded(stores, F: int, W: int, H: int, rays_dir_cache, kf_meta: torch.Tensor, n_frames: int, n_px: int,
                       n_cam2surf: int, n_bins: int, surface_eps: float, stop_eps: float, min_bound: float = 0.0,
                       obj_center: float = 0.0, seed: Optional[int] = None, draw: Optional[int] = None,
                       obj_index: int = 0, want_pts: bool = False, record: bool = False,
                       kf_ids: Optional[torch.Tensor] = None, partfeat=None) -> Dict[str, Optional[torch.Tensor]]:
    """The sampler with its random numbers generated in the kernels (Philox keyed on seed / draw / object / ray / bin;
    nothing random is stored).  stores: a keyframe table [K, 4] (ops.keyframe_table -> stacked call, tensors
    [K, ...]) or the four store tensors of ONE object (tensors without the leading K).  kf_meta int32 [K, 4] | [4]:
    n_keyframes, the slots of the latest two keyframes (-1 while n_keyframes <= 2), the object's random-stream id
    (without kf_meta: obj_index (+ k)); kf_ids overrides the seeded
    keyframe choice.  want_pts = False returns origins / dirs / z -- the pts == NULL form of ops.train_step, the
    [.., n, S, 3] point tensor is never written; record = True adds the drawn keyframes `kf` and pixels `px`.
    partfeat = (global_partfeat, use_frame, stride, part_down): the part-level feature of every ray (vmap.py:437-452)
    is gathered by the same launch -> key "partfeat" [.., n, C].
    Returns a dict: rgb u8, depth, valid (bool), labels u8, z, pts | origins + dirs, kf, px."""
    stacked = torch.is_tensor(stores)
    rays_dir_cache = _req(rays_dir_cache, torch.float32, "rays_dir_cache")
    kf_meta = _req(kf_meta, torch.int32, "kf_meta") if kf_meta is not None else None
    kf_ids = _req(kf_ids, torch.int64, "kf_ids") if kf_ids is not None else None
    if stacked:
        table = _req(stores, torch.int64, "table")
        K, dev = table.shape[0], table.device
        if (kf_meta is not None and kf_meta.shape != (K, 4)) or (kf_ids is not None and kf_ids.shape != (K, n_frames)):
            raise ObjnerfError("sample_rays_seeded: kf_meta / kf_ids do not match the table")
        ptrs = [None] * 4
    else:
        rgbs, depth, t_wc, bbox = stores
        ptrs = [_ptr(_req(rgbs, torch.uint8, "rgbs_batch")), _ptr(_req(depth, torch.float32, "depth_batch")),
                _ptr(_req(t_wc, torch.float32, "t_wc_batch")), _ptr(_req(bbox, torch.float32, "bbox"))]
        K, dev = 1, rgbs.device
    if kf_meta is None and kf_ids is None:
        raise ObjnerfError("sample_rays_seeded: kf_meta or kf_ids")
    o = _sample_common(K, n_frames, n_px, n_cam2surf, n_bins, dev, want_pts, record, stacked)
    a = SampleArgs(F, W, H, n_frames, n_px, n_cam2surf, n_bins, int(obj_index), surface_eps, stop_eps, min_bound,
                   obj_center, *ptrs, _ptr(rays_dir_cache), _ptr(kf_ids), None, None, None, None,
                   _ptr(o["rgb"]), _ptr(o["depth"]), _ptr(o["valid"]), _ptr(o["labels"]), _ptr(o["z"]), _ptr(o["pts"]),
                   _ptr(o["ws"]), _seed_of(seed), (_next_offset() if draw is None else int(draw)) & 0x1FFFFFFF, 0,
                   _ptr(kf_meta), _ptr(o["kf"]), _ptr(o["px"]), _ptr(o["origins"]), _ptr(o["dirs"]))
    o["partfeat"] = _partfeat_fields(a, partfeat, K, F, n_frames * n_px, dev, stacked)
    if stacked:
        check(lib().objnerf_sample_rays_stacked(C.byref(a), K, _ptr(table), _stream()), "objnerf_sample_rays_stacked")
    else:
        check(lib().objnerf_sample_rays(C.byref(a), _stream()), "objnerf_sample_rays")
    if kf_ids is not None and record:
        o["kf"] = kf_ids
    o["valid"] = o["valid"].bool()
    del o["ws"]
    return o


def rays_dirs(W: int, H: int, fx: float, fy: float, cx: float, cy: float, device) -> torch.Tensor:
    out = torch.empty(W, H, 3, device=device)
    check(lib().objnerf_rays_dirs(W, H, fx, fy, cx, cy, _ptr(out), _stream()), "objnerf_rays_dirs")
    return out


def sample_rays(rgbs_batch, depth_batch, t_wc_batch, bbox, rays_dir_cache, kf_ids, u_w, u_h, u, g,
                n_cam2surf: int, n_bins: int, surface_eps: float, stop_eps: float, min_bound: float = 0.0,
                obj_center: float = 0.0, partfeat=None):
    """sceneObject.get_training_samples + sample_3d_points with injected draws (vmap.py:386-554).
    partfeat = (global_partfeat, use_frame, stride, part_down): also returns sampled_partfeat [n_frames, n_px, C]
    (vmap.py:437-452), gathered by the same launch."""
    rgbs_batch = _req(rgbs_batch, torch.uint8, "rgbs_batch")
    depth_batch = _req(depth_batch, torch.float32, "depth_batch")
    t_wc_batch = _req(t_wc_batch, torch.float32, "t_wc_batch")
    bbox = _req(bbox, torch.float32, "bbox")
    rays_dir_cache = _req(rays_dir_cache, torch.float32, "rays_dir_cache")
    kf_ids = _req(kf_ids, torch.int64, "kf_ids")
    u_w = _req(u_w, torch.float32, "u_w")
    u_h = _req(u_h, torch.float32, "u_h")
    u = _req(u, torch.float32, "u")
    g = _req(g, torch.float32, "g")
    F, W, H = rgbs_batch.shape[0], rgbs_batch.shape[1], rgbs_batch.shape[2]
    n_frames, n_px = u_w.shape
    n = n_frames * n_px
    S = n_cam2surf + n_bins
    dev = rgbs_batch.device
    out_rgb = torch.empty(n_frames, n_px, 3, dtype=torch.uint8, device=dev)
    out_depth = torch.empty(n_frames, n_px, device=dev)
    out_valid = torch.empty(n, dtype=torch.uint8, device=dev)
    out_labels = torch.empty(n, dtype=torch.uint8, device=dev)
    out_z = torch.empty(n_frames, n_px, S, device=dev)
    out_pts = torch.empty(n_frames, n_px, S, 3, device=dev)
    ws = torch.empty(1 + 6 * n, device=dev)
    a = SampleArgs(F, W, H, n_frames, n_px, n_cam2surf, n_bins, 0, surface_eps, stop_eps, min_bound, obj_center,
                   _ptr(rgbs_batch), _ptr(depth_batch), _ptr(t_wc_batch), _ptr(bbox), _ptr(rays_dir_cache),
                   _ptr(kf_ids), _ptr(u_w), _ptr(u_h), _ptr(u), _ptr(g), _ptr(out_rgb), _ptr(out_depth),
                   _ptr(out_valid), _ptr(out_labels), _ptr(out_z), _ptr(out_pts), _ptr(ws), 0, 0, 0, None, None, None,
                   None, None)
    pf = _partfeat_fields(a, partfeat, 1, F, n, dev, False)
    check(lib().objnerf_sample_rays(C.byref(a), _stream()), "objnerf_sample_rays")
    r = (out_rgb, out_depth, out_valid.bool(), out_labels, out_pts, out_z)
    return r if partfeat is None else r + (pf.reshape(n_frames, n_px, -1),)


def sample_points(sampled_rgbs, sampled_depth, origins, dirs_w, n_cam2surf: int, n_bins: int, surface_eps: float,
                  stop_eps: float, min_bound: float = 0.0, obj_center: float = 0.0, u=None, g=None, seed=None,
                  obj_index: int = 0):
    """sceneObject.sample_3d_points alone (vmap.py:456-554, objnerf_sample_points): the pixels are already gathered.
    sampled_rgbs u8 [n_frames, n_px, 4] (rgb + state), sampled_depth [n_frames, n_px], origins [n_frames, 3], dirs_w
    [n_frames, n_px, 3].  u [n, N + M] / g [n, M]: the reference's draws injected (both or neither; neither = Philox
    draws under `seed` and a per-call counter).  -> (rgb u8, depth, valid bool [n], labels u8 [n], pts, z)."""
    sampled_rgbs = _req(sampled_rgbs, torch.uint8, "sampled_rgbs")
    sampled_depth = _req(sampled_depth, torch.float32, "sampled_depth")
    origins = _req(origins, torch.float32, "origins")
    dirs_w = _req(dirs_w, torch.float32, "dirs_w")
    n_frames, n_px = sampled_depth.shape
    if tuple(sampled_rgbs.shape) != (n_frames, n_px, 4) or tuple(origins.shape) != (n_frames, 3) or \
            tuple(dirs_w.shape) != (n_frames, n_px, 3):
        raise ObjnerfError("sample_points: sampled_rgbs [F,P,4], sampled_depth [F,P], origins [F,3], dirs_w [F,P,3]")
    if (u is None) != (g is None):
        raise ObjnerfError("sample_points: inject both u and g or neither")
    n, S = n_frames * n_px, n_cam2surf + n_bins
    dev = sampled_rgbs.device
    if u is not None:
        u = _req(u, torch.float32, "u")
        g = _req(g, torch.float32, "g")
        draw = 0
    else:
        draw = _next_offset() & 0x1FFFFFFF
    out_rgb = torch.empty(n_frames, n_px, 3, dtype=torch.uint8, device=dev)
    out_depth = torch.empty(n_frames, n_px, device=dev)
    out_valid = torch.empty(n, dtype=torch.uint8, device=dev)
    out_labels = torch.empty(n, dtype=torch.uint8, device=dev)
    out_z = torch.empty(n_frames, n_px, S, device=dev)
    out_pts = torch.empty(n_frames, n_px, S, 3, device=dev)
    ws = torch.empty(1 + 6 * n, device=dev)
    a = SampleArgs(0, 0, 0, n_frames, n_px, n_cam2surf, n_bins, int(obj_index), surface_eps, stop_eps, min_bound,
                   obj_center, None, None, None, None, None, None, None, None, _ptr(u), _ptr(g), _ptr(out_rgb),
                   _ptr(out_depth), _ptr(out_valid), _ptr(out_labels), _ptr(out_z), _ptr(out_pts), _ptr(ws),
                   _seed_of(seed), draw, 0, None, None, None, None, None)
    check(lib().objnerf_sample_points(C.byref(a), _ptr(sampled_rgbs), _ptr(sampled_depth), _ptr(origins), _ptr(dirs_w),
                                      _stream()), "objnerf_sample_points")
    return out_rgb, out_depth, out_valid.bool(), out_labels, out_pts, out_z


# ------------------------------------------------------------------------------------------------
# helper functions of the reference's call surface (objnerf_helpers.hip)
# ------------------------------------------------------------------------------------------------
def render(termination: torch.Tensor, vals: torch.Tensor) -> torch.Tensor:
    """render_rays.render over the last axis of termination [..., S] with vals [..., S] or [..., S, C] (objnerf_render)."""
    termination = _req(termination, torch.float32, "termination")
    S = termination.shape[-1]
    scalar = vals.dim() == termination.dim()
    Cdim = 1 if scalar else vals.shape[-1]
    vals = _req(vals, torch.float32, "vals")
    if tuple(vals.shape[:termination.dim()]) != tuple(termination.shape):
        raise ObjnerfError("render: vals must be termination's shape (+ a channel axis)")
    n = termination.numel() // S
    out = torch.empty(termination.shape[:-1] + (() if scalar else (Cdim,)), device=termination.device)
    if n:
        check(lib().objnerf_render(n, S, Cdim, _ptr(termination), _ptr(vals), _ptr(out), _stream()), "objnerf_render")
    return out


def render_loss(render: torch.Tensor, gt: torch.Tensor, mode: int, normalise: bool = False) -> torch.Tensor:
    """render_rays.render_loss: mode 0 L1, 1 L2 (elementwise, any shape); 2 cos (over the last axis)."""
    render = _req(render, torch.float32, "render")
    gt = _req(gt.expand_as(render) if gt.shape != render.shape else gt, torch.float32, "gt")
    if mode == 2:
        Cdim = render.shape[-1]
        out = torch.empty(render.shape[:-1], device=render.device)
        n = out.numel()
    else:
        Cdim, out = 1, torch.empty_like(render)
        n = out.numel()
    if n:
        check(lib().objnerf_render_loss(n, Cdim, mode, int(normalise), _ptr(render), _ptr(gt), _ptr(out), _stream()),
              "objnerf_render_loss")
    return out


def reduce_batch_loss(loss_mat: torch.Tensor, var, mask: torch.Tensor, l2: bool, avg: bool):
    """render_rays.reduce_batch_loss on [K,R] tensors -> (out [K] or [K,R], status int32[1])."""
    loss_mat = _req(loss_mat, torch.float32, "loss_mat")
    K, R = loss_mat.shape
    var = _req(var, torch.float32, "var") if var is not None else None
    mask = _req(mask.to(torch.uint8), torch.uint8, "mask")
    dev = loss_mat.device
    out = torch.empty(K if avg else (K, R), device=dev) if avg else torch.empty(K, R, device=dev)
    ws = torch.empty(K + 1, dtype=torch.int32, device=dev)
    status = torch.zeros(1, dtype=torch.int32, device=dev)
    check(lib().objnerf_reduce_batch_loss(K, R, _ptr(loss_mat), _ptr(var), _ptr(mask), int(l2), int(avg), _ptr(ws), _ptr(out),
                                          _ptr(status), _stream()), "objnerf_reduce_batch_loss")
    return out, status


def make_grid(dim: int, lo: float, hi: float, scale, transform, device) -> torch.Tensor:
    out = torch.empty(dim, dim, dim, 3, device=device)
    scale = _req(scale.to(device).float().reshape(3), torch.float32, "scale") if scale is not None else None
    transform = _req(transform.to(device).float().reshape(4, 4), torch.float32, "transform") if transform is not None else None
    check(lib().objnerf_make_grid(dim, float(lo), float(hi), _ptr(scale), _ptr(transform), _ptr(out), _stream()),
          "objnerf_make_grid")
    return out


def ray_box(origins, directions, bounds_min, bounds_max):
    origins = _req(origins, torch.float32, "origins")
    directions = _req(directions, torch.float32, "directions")
    n, dev = origins.shape[0], origins.device
    bmin = _req(torch.as_tensor(bounds_min, dtype=torch.float32, device=dev).reshape(3), torch.float32, "bounds_min")
    bmax = _req(torch.as_tensor(bounds_max, dtype=torch.float32, device=dev).reshape(3), torch.float32, "bounds_max")
    near, far = torch.empty(n, device=dev), torch.empty(n, device=dev)
    hit = torch.empty(n, dtype=torch.uint8, device=dev)
    check(lib().objnerf_ray_box(n, _ptr(origins), _ptr(directions), _ptr(bmin), _ptr(bmax), _ptr(near), _ptr(far), _ptr(hit),
                                _stream()), "objnerf_ray_box")
    return near, far, hit.bool()


def dirs_w(T_WC: torch.Tensor, dirs_C: torch.Tensor) -> torch.Tensor:
    T_WC = _req(T_WC, torch.float32, "T_WC")
    dirs_C = _req(dirs_C, torch.float32, "dirs_C")
    F = T_WC.shape[0]
    P = dirs_C.numel() // (3 * F)
    out = torch.empty_like(dirs_C)
    check(lib().objnerf_dirs_w(F, P, _ptr(T_WC), _ptr(dirs_C), _ptr(out), _stream()), "objnerf_dirs_w")
    return out


_draw_offset = [0]


def _next_offset() -> int:
    _draw_offset[0] += 1
    return _draw_offset[0]


def stratified_bins(lo, hi, n_bins: int, n_rays: int, device, u: Optional[torch.Tensor] = None,
                    seed: Optional[int] = None, draw: Optional[int] = None) -> torch.Tensor:
    """lo / hi: python scalars or [n_rays] tensors.  u: injected uniforms [n_rays, n_bins]; otherwise the counter-based
    generator under `seed` (default: torch's initial seed) and the call counter `draw` (default: advances per call)."""
    lo_t = _req(lo.to(device), torch.float32, "min_depth") if torch.is_tensor(lo) else None
    hi_t = _req(hi.to(device), torch.float32, "max_depth") if torch.is_tensor(hi) else None
    u = _req(u, torch.float32, "u") if u is not None else None
    out = torch.empty(n_rays, n_bins, device=device)
    sd = torch.initial_seed() if seed is None else seed
    check(lib().objnerf_stratified_bins(n_rays, n_bins, _ptr(lo_t), 0.0 if lo_t is not None else float(lo), _ptr(hi_t),
                                        0.0 if hi_t is not None else float(hi), _ptr(u), sd & (2 ** 64 - 1),
                                        _next_offset() if draw is None else int(draw),
                                        _ptr(out), _stream()), "objnerf_stratified_bins")
    return out


def normal_bins(depth: torch.Tensor, n_bins: int, delta: float, g: Optional[torch.Tensor] = None,
                seed: Optional[int] = None, draw: Optional[int] = None) -> torch.Tensor:
    depth = _req(depth, torch.float32, "depth")
    n_rays = depth.shape[0]
    g = _req(g, torch.float32, "g") if g is not None else None
    out = torch.empty(n_rays, n_bins, device=depth.device)
    sd = torch.initial_seed() if seed is None else seed
    check(lib().objnerf_normal_bins(n_rays, n_bins, _ptr(depth), float(delta), _ptr(g), sd & (2 ** 64 - 1),
                                    _next_offset() if draw is None else int(draw), _ptr(out), _stream()),
          "objnerf_normal_bins")
    return out
