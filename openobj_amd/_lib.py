"""ctypes binding of libobjnerf_hip.so (include/objnerf_hip.h).

The library is built in-tree (openobj_amd/csrc/libobjnerf_hip.so, `make -C openobj_amd/csrc` or
`__graft_entry__.build()`).  There is NO fallback: if it is missing, or a call returns an error
code, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

# torch must be imported BEFORE libobjnerf_hip.so is dlopen'ed: the torch wheel bundles its own
# libamdhip64.so.7 and both must share ONE HIP runtime in the process (streams, device pointers).
# Loading ours first would bind /opt/rocm's runtime and kernel launches on torch's streams fail.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OBJNERF_LIB") or os.path.join(_HERE, "csrc", "libobjnerf_hip.so")   # OBJNERF_LIB: diagnostic builds

OBJNERF_N_TENSORS = 19
ABI_VERSION = 7


class ObjnerfError(RuntimeError):
    pass


class Net(C.Structure):
    _fields_ = [("hidden", C.c_int32), ("feat_dim", C.c_int32), ("n_freqs", C.c_int32),
                ("reserved", C.c_int32)]


class SampleArgs(C.Structure):
    _fields_ = [("F", C.c_int32), ("W", C.c_int32), ("H", C.c_int32), ("n_frames", C.c_int32),
                ("n_px", C.c_int32), ("n_cam2surf", C.c_int32), ("n_bins", C.c_int32),
                ("obj_index", C.c_int32),
                ("surface_eps", C.c_float), ("stop_eps", C.c_float), ("min_bound", C.c_float),
                ("obj_center", C.c_float),
                ("rgbs", C.c_void_p), ("depth", C.c_void_p), ("t_wc", C.c_void_p), ("bbox", C.c_void_p),
                ("rays_dir_cache", C.c_void_p),
                ("kf_ids", C.c_void_p), ("u_w", C.c_void_p), ("u_h", C.c_void_p), ("u", C.c_void_p),
                ("g", C.c_void_p),
                ("out_rgb", C.c_void_p), ("out_depth", C.c_void_p), ("out_valid", C.c_void_p),
                ("out_labels", C.c_void_p), ("out_z", C.c_void_p), ("out_pts", C.c_void_p),
                ("max_depth_ws", C.c_void_p),
                ("seed", C.c_uint64), ("draw", C.c_uint32), ("reserved", C.c_uint32),
                ("kf_meta", C.c_void_p), ("out_kf", C.c_void_p), ("out_px", C.c_void_p),
                ("out_origins", C.c_void_p), ("out_dirs", C.c_void_p),
                # ABI 6: the part-feature gather (vmap.py:437-452)
                ("global_partfeat", C.c_void_p), ("use_frame", C.c_void_p), ("out_partfeat", C.c_void_p),
                ("pf_frames", C.c_int32), ("pf_w", C.c_int32), ("pf_h", C.c_int32), ("pf_c", C.c_int32),
                ("pf_stride", C.c_int32), ("part_down", C.c_float)]


class IngestItem(C.Structure):
    _fields_ = [("rgbs", C.c_void_p), ("depth", C.c_void_p), ("t_wc", C.c_void_p), ("bbox", C.c_void_p),
                ("slot", C.c_int32), ("obj_id", C.c_int32), ("box", C.c_float * 4)]


class LossArgs(C.Structure):
    _fields_ = [("K", C.c_int32), ("R", C.c_int32), ("S", C.c_int32), ("C", C.c_int32),
                ("color_scaling", C.c_float), ("opacity_scaling", C.c_float),
                ("feat_scaling", C.c_float), ("reserved", C.c_float),
                ("alpha", C.c_void_p), ("color", C.c_void_p), ("z", C.c_void_p), ("gt_depth", C.c_void_p),
                ("gt_rgb", C.c_void_p), ("labels", C.c_void_p), ("pred_feat", C.c_void_p),
                ("gt_feat", C.c_void_p), ("flags_in", C.c_void_p), ("counts_in", C.c_void_p),
                ("loss_terms", C.c_void_p), ("total", C.c_void_p), ("d_alpha", C.c_void_p),
                ("d_color", C.c_void_p), ("d_pred_feat", C.c_void_p),
                ("counts", C.c_void_p), ("status", C.c_void_p)]


TRAIN_BF16 = 1       # objnerf_train_args.mode bit (OBJNERF_TRAIN_BF16)
TRAIN_SELF_COUNTS = 8    # (OBJNERF_TRAIN_SELF_COUNTS)


class AdamWArgs(C.Structure):
    _fields_ = [("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("group_steps", C.c_void_p),
                ("bank", C.c_int32), ("reserved", C.c_int32),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("reserved_f", C.c_float)]


class TrainArgs(C.Structure):
    _fields_ = [("K", C.c_int32), ("R", C.c_int32), ("S", C.c_int32), ("mode", C.c_int32),
                ("color_scaling", C.c_float), ("opacity_scaling", C.c_float),
                ("feat_scaling", C.c_float), ("obj_center", C.c_float),
                ("params", C.c_void_p), ("p_stride", C.c_int64), ("scale", C.c_void_p),
                ("pts", C.c_void_p), ("origins", C.c_void_p), ("dirs", C.c_void_p), ("z", C.c_void_p),
                ("gt_depth", C.c_void_p), ("gt_rgb", C.c_void_p), ("labels", C.c_void_p),
                ("gt_feat", C.c_void_p),
                ("counts", C.c_void_p), ("flags", C.c_void_p),
                ("grads", C.c_void_p), ("loss_terms", C.c_void_p), ("status", C.c_void_p),
                ("workspace", C.c_void_p), ("workspace_bytes", C.c_size_t),
                ("relu_masks", C.c_void_p), ("context", C.c_void_p), ("emb_debug", C.c_void_p),
                ("optim", C.c_void_p)]


# name -> (restype, argtypes); every symbol include/objnerf_hip.h declares
SIGNATURES = {
    "objnerf_abi_version": (C.c_int, []),
    "objnerf_context_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "objnerf_context_destroy": (C.c_int, [C.c_void_p]),
    "objnerf_param_layout": (C.c_int64, [C.POINTER(Net), C.POINTER(C.c_int64)]),
    "objnerf_rays_dirs": (C.c_int, [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_float,
                                    C.c_void_p, C.c_void_p]),
    "objnerf_sample_rays": (C.c_int, [C.POINTER(SampleArgs), C.c_void_p]),
    "objnerf_sample_rays_stacked": (C.c_int, [C.POINTER(SampleArgs), C.c_int32, C.c_void_p, C.c_void_p]),
    "objnerf_sample_points": (C.c_int, [C.POINTER(SampleArgs), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_ingest_frame": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "objnerf_box_rays": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_box_points": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_render_fwd": (C.c_int, [C.POINTER(Net), C.c_int64, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "objnerf_eval_points": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "objnerf_eval_workspace_bytes": (C.c_size_t, [C.POINTER(Net), C.c_int32, C.c_int64]),
    "objnerf_eval_points_ws": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "objnerf_mlp_forward": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "objnerf_mlp_forward_ws": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "objnerf_embed": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_mlp_backward_workspace_bytes": (C.c_size_t, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_int32]),
    "objnerf_mlp_backward_ws": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                          C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_size_t, C.c_void_p]),
    "objnerf_embed_bwd": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_occupancy": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_render": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_render_loss": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "objnerf_reduce_batch_loss": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_make_grid": (C.c_int, [C.c_int32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_ray_box": (C.c_int, [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "objnerf_dirs_w": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_stratified_bins": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p,
                                          C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "objnerf_normal_bins": (C.c_int, [C.c_int64, C.c_int32, C.c_void_p, C.c_float, C.c_void_p, C.c_uint64, C.c_uint64,
                                      C.c_void_p, C.c_void_p]),
    "objnerf_mfma_peak": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "objnerf_composite": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p]),
    "objnerf_feature_head": (C.c_int, [C.POINTER(Net), C.c_int32, C.c_int64, C.c_void_p, C.c_int64,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "objnerf_step_batch_loss": (C.c_int, [C.POINTER(LossArgs), C.c_void_p]),
    "objnerf_label_counts": (C.c_int, [C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.c_void_p]),
    "objnerf_train_workspace_bytes": (C.c_size_t, [C.POINTER(Net), C.c_int32, C.c_int32, C.c_int32,
                                                   C.c_int32]),
    "objnerf_train_step": (C.c_int, [C.POINTER(Net), C.POINTER(TrainArgs), C.c_void_p]),
    "objnerf_adamw_step": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, C.c_void_p]),
    "objnerf_adamw_step_flags": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_int64,
                                           C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the library; raises ObjnerfError if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ObjnerfError(
                f"{LIB_PATH} is missing: build it with `make -C openobj_amd/csrc` "
                "(or __graft_entry__.build()).  There is no CPU / PyTorch fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if l.objnerf_abi_version() != ABI_VERSION:
            raise ObjnerfError("libobjnerf_hip.so ABI version mismatch")
        _lib = l
    return _lib


_ERR = {-22: "EINVAL (bad shape / null pointer)", -95: "ENOTSUP (shape not built)",
        -5: "ELAUNCH (HIP launch failed)"}


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise ObjnerfError(f"{what} failed: {rc} {_ERR.get(rc, '')}")


def param_layout(hidden: int, feat_dim: int = 512, n_freqs: int = 6):
    """-> (offsets[20], p_stride)"""
    net = Net(hidden, feat_dim, n_freqs, 0)
    offs = (C.c_int64 * (OBJNERF_N_TENSORS + 1))()
    ps = lib().objnerf_param_layout(C.byref(net), offs)
    if ps < 0:
        raise ObjnerfError("objnerf_param_layout failed")
    return list(offs), int(ps)
