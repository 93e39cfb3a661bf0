"""UniDirsEmbed -- the reference's positional encoding (embedding.py:4-55): sin on 21 learnable
icosahedron directions x (max_deg - min_deg + 1) octaves, x/scale prepended.  `B_layer.weight` lives in
the parameter arena; forward runs objnerf_embed."""
import torch

from . import ops
from .autograd import EmbedFunction
from .init import icosa_dirs


class UniDirsEmbed(torch.nn.Module):
    def __init__(self, min_deg=0, max_deg=2, scale=2., device=None, _arena=None):
        super().__init__()
        if min_deg != 0:
            raise NotImplementedError("min_deg != 0 is never used by the reference")
        self.min_deg, self.max_deg = min_deg, max_deg
        self.n_freqs = max_deg - min_deg + 1
        self.tensor_scale = torch.tensor(scale, requires_grad=False)
        if device is None:
            device = "cuda" if torch.cuda.is_available() else "cpu"
        self._arena = _arena if _arena is not None else ops.ParamArena(1, ops.NetShape(32, 512, self.n_freqs), device)
        if self._arena.net.n_freqs != self.n_freqs:
            raise ValueError("arena built for a different number of octaves")
        self.B_layer = torch.nn.Linear(3, 21, bias=False)
        bview = self._arena.views()[18][0]
        with torch.no_grad():
            bview.copy_(icosa_dirs())                       # embedding.py:15-40
        self.B_layer.weight = torch.nn.Parameter(bview)
        self._arena.set_scale(float(scale))
        frequency_bands = 2.0 ** torch.linspace(self.min_deg, self.max_deg, self.n_freqs)
        self.register_buffer("frequency_bands", frequency_bands, persistent=False)
        self.register_buffer("scale", self.tensor_scale, persistent=True)

    def forward(self, x):
        """x [...,3] -> [..., 3 + 21 * n_freqs] (embedding.py:46-55)."""
        lead = x.shape[:-1]
        self._arena.set_scale(float(self.scale))
        emb = EmbedFunction.apply(self._arena, False, x.reshape(1, -1, 3).contiguous(), self.B_layer.weight)
        return emb.reshape(*lead, emb.shape[-1])
