"""Novel-view rendering of a whole scene: every object is rendered inside its own oriented box
(sceneObject.render_2D_syn) and the per-object images are merged by depth -- the loop of train.py:550-612
(`cfg.if_render`).  Images are stored transposed, [W, H, ...], like everything else in the reference."""
from typing import Dict, Iterable, Optional

import numpy as np


class ViewBuffers:
    """rendered_rgb_image / rendered_depth_image / rendered_maskid_image of train.py:558-566."""

    def __init__(self, W: int, H: int):
        self.rgb = np.zeros((W, H, 3), dtype=np.uint8)
        self.maskid = np.zeros((W, H), dtype=np.int32)
        self.depth = np.ones((W, H), dtype=np.float32) * 100        # "set the depth image large first" (:566)

    def merge(self, obj_mask: np.ndarray, render_depth: np.ndarray, render_color: np.ndarray, class_id: int,
              is_background: bool) -> np.ndarray:
        """z-buffer test of one object's render against what is already there (train.py:581-598).  Background
        objects paint colour but do not write depth, so that they never hide a foreground object (:596-597).
        Returns the boolean image of the pixels this object won."""
        this_depth = np.ones_like(self.depth) * 100
        this_rgb = np.zeros_like(self.rgb)
        this_depth[obj_mask] = render_depth
        this_rgb[obj_mask] = render_color
        ok = self.depth > this_depth
        self.rgb[ok] = this_rgb[ok]
        self.maskid[ok] = class_id
        if not is_background:
            self.depth[ok] = this_depth[ok]
        return ok


def render_view(vis_dict: Dict[int, object], T_WC: np.ndarray, rays_dir, intrinsic_open3d=None,
                bg_ids: Iterable[int] = (0,), class_of: Optional[Dict[int, int]] = None, W: Optional[int] = None,
                H: Optional[int] = None, draws: Optional[Dict[int, object]] = None) -> ViewBuffers:
    """Render every object of `vis_dict` (obj_id -> sceneObject, dict order as train.py:570) from pose T_WC and
    merge.  class_of maps obj_id -> the id written into the mask-id image (mapping_class of train.py:592)."""
    first = next(iter(vis_dict.values()))
    W = W or first.trainer.W_vis
    H = H or first.trainer.H_vis
    bg_ids = set(bg_ids)
    buf = ViewBuffers(W, H)
    for obj_id, obj_k in vis_dict.items():
        res = obj_k.render_2D_syn(T_WC, intrinsic_open3d, rays_dir, chunk_size=3000, do_fine=False,
                                  draws=None if draws is None else draws.get(obj_id))
        if res[1] is None:
            continue
        obj_mask, render_depth, render_color = res[0], res[1], res[2]
        buf.merge(obj_mask, render_depth, render_color, (class_of or {}).get(obj_id, obj_id), obj_id in bg_ids)
    return buf
