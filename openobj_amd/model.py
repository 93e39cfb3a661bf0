"""OccupancyMap -- the reference's per-object MLP (model.py:17-103) with its parameters living in the
MI355X parameter arena and its forward running in libobjnerf_hip.so.

Same constructor signature, same sub-module names (so `state_dict()` keys are the reference's:
in_layer.0.*, mid1.0.0.*, cat_layer.0.*, mid2.0.0.*, out_alpha.*, color_linear.0.*, out_color.*,
clip_linear.0.*, out_clip.*), same initialisation stream (nn.Linear construction order, then
`apply(init_weights)` = xavier-normal weights).
"""
import torch
import torch.nn as nn

from . import ops
from .autograd import MlpFunction


def init_weights(m, init_fn=torch.nn.init.xavier_normal_):
    if type(m) == torch.nn.Linear:        # model.py:4-6
        # The reference initialises on the CPU (then .to(device)); draw from the CPU generator as well so
        # that a seed yields the same weights, then write into the (possibly device-resident) arena view.
        tmp = torch.empty(m.weight.shape, dtype=m.weight.dtype)
        init_fn(tmp)
        with torch.no_grad():
            m.weight.copy_(tmp)


def fc_block(in_f, out_f):
    return torch.nn.Sequential(torch.nn.Linear(in_f, out_f), torch.nn.ReLU(out_f))   # model.py:9-13


def _repoint(linear: nn.Linear, w_view: torch.Tensor, b_view):
    """Move a freshly constructed nn.Linear's values into the arena and make its Parameters views of it."""
    with torch.no_grad():
        w_view.copy_(linear.weight)
        if b_view is not None:
            b_view.copy_(linear.bias)
    linear.weight = nn.Parameter(w_view)
    if b_view is not None:
        linear.bias = nn.Parameter(b_view)


class OccupancyMap(torch.nn.Module):
    def __init__(self, emb_size1, emb_size2, hidden_size=256, do_color=True, do_clip=True, clip_size=512,
                 hidden_layers_block=1, device=None, _arena=None):
        super().__init__()
        if hidden_layers_block != 1 or not (do_color and do_clip):
            raise NotImplementedError("only the configuration Trainer builds (trainer.py:36-44) is supported")
        if (emb_size1, emb_size2) != (ops.EMB1, ops.EMB2):
            raise NotImplementedError("embedding split must be 87/42 (trainer.py:20-21)")
        self.do_color, self.do_clip = do_color, do_clip
        self.embedding_size1, self.embedding_size2 = emb_size1, emb_size2
        # nn.Linear construction order of model.py:29-56 (same RNG consumption as the reference)
        self.in_layer = fc_block(emb_size1, hidden_size)
        self.mid1 = torch.nn.Sequential(*[fc_block(hidden_size, hidden_size) for _ in range(hidden_layers_block)])
        self.cat_layer = fc_block(hidden_size + emb_size1, hidden_size)
        self.mid2 = torch.nn.Sequential(*[fc_block(hidden_size, hidden_size) for _ in range(hidden_layers_block)])
        self.out_alpha = torch.nn.Linear(hidden_size, 1)
        self.color_linear = fc_block(emb_size2 + hidden_size, hidden_size)
        self.out_color = torch.nn.Linear(hidden_size, 3)
        self.clip_linear = fc_block(emb_size2 + hidden_size, hidden_size)
        self.out_clip = torch.nn.Linear(hidden_size, clip_size)
        self.sigmoid = torch.sigmoid
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self._arena = _arena if _arena is not None else ops.ParamArena(1, ops.NetShape(hidden_size, clip_size), device)
        v = [t[0] for t in self._arena.views()]
        lins = [self.in_layer[0], self.mid1[0][0], self.cat_layer[0], self.mid2[0][0], self.out_alpha,
                self.color_linear[0], self.out_color, self.clip_linear[0], self.out_clip]
        for i, lin in enumerate(lins):
            _repoint(lin, v[2 * i], v[2 * i + 1])

    def forward(self, x, noise_std=None, do_alpha=True, do_color=True, do_cat=True, do_clip=True):
        """x [..., 129] (an embedding) -> (alpha [...,1], color [...,3], clip [...,C]); model.py:61-103.
        Differentiable (autograd.MlpFunction -> objnerf_mlp_backward_ws) w.r.t. the parameters and x; the fast training
        path is training_strategy == "hip"."""
        if not do_cat:
            raise NotImplementedError("do_cat=False is never used by the reference")
        lead = x.shape[:-1]
        emb = x.reshape(1, -1, x.shape[-1]).contiguous()
        want_clip = self.do_clip and do_clip
        alpha, color, clip = MlpFunction.apply(self._arena, want_clip, False, emb, *self.parameters())
        alpha = alpha.reshape(*lead, 1)
        if noise_std is not None:                       # model.py:83-85: raw + noise, then * 10
            alpha = alpha + 10.0 * noise_std * torch.randn(alpha.shape, device=alpha.device)
        color = color.reshape(*lead, 3)
        clip = clip.reshape(*lead, -1) if want_clip else None
        return (alpha if do_alpha else None), (color if (self.do_color and do_color) else None), clip
