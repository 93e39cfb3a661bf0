"""Parameter initialisation of the object networks (host side).

Reference: trainer.py:36-44 builds model.OccupancyMap and applies model.init_weights
(xavier_normal_ on every nn.Linear weight, model.py:4-6); biases keep nn.Linear's default
U(-1/sqrt(fan_in), 1/sqrt(fan_in)); UniDirsEmbed.B_layer.weight starts at the 21 icosahedron
directions (embedding.py:15-40).
"""
from __future__ import annotations

import math
from typing import List, Optional

import torch

from .ops import tensor_shapes

# embedding.py:15-37
ICOSA_DIRS = (
    0.8506508, 0, 0.5257311, 0.809017, 0.5, 0.309017, 0.5257311, 0.8506508, 0, 1, 0, 0,
    0.809017, 0.5, -0.309017, 0.8506508, 0, -0.5257311, 0.309017, 0.809017, -0.5,
    0, 0.5257311, -0.8506508, 0.5, 0.309017, -0.809017, 0, 1, 0, -0.5257311, 0.8506508, 0,
    -0.309017, 0.809017, -0.5, 0, 0.5257311, 0.8506508, -0.309017, 0.809017, 0.5,
    0.309017, 0.809017, 0.5, 0.5, 0.309017, 0.809017, 0.5, -0.309017, 0.809017, 0, 0, 1,
    -0.5, 0.309017, 0.809017, -0.809017, 0.5, 0.309017, -0.809017, 0.5, -0.309017,
)


def icosa_dirs() -> torch.Tensor:
    return torch.tensor(ICOSA_DIRS, dtype=torch.float32).reshape(21, 3)


def init_object_tensors(hidden: int, feat_dim: int = 512,
                        generator: Optional[torch.Generator] = None) -> List[torch.Tensor]:
    """The 19 tensors of one object (18 OccupancyMap parameters in parameters() order + B)."""
    shapes = tensor_shapes(hidden, feat_dim)
    out: List[torch.Tensor] = []
    for i, shp in enumerate(shapes[:18]):
        t = torch.empty(shp, dtype=torch.float32)
        if len(shp) == 2:
            fan_out, fan_in = shp
            t.normal_(0.0, math.sqrt(2.0 / float(fan_in + fan_out)), generator=generator)
        else:
            fan_in = shapes[i - 1][1]
            bound = 1.0 / math.sqrt(fan_in)
            t.uniform_(-bound, bound, generator=generator)
        out.append(t)
    out.append(icosa_dirs())
    return out


def init_stacked(K: int, hidden: int, feat_dim: int = 512, seed: int = 0) -> List[torch.Tensor]:
    gen = torch.Generator().manual_seed(seed)
    objs = [init_object_tensors(hidden, feat_dim, gen) for _ in range(K)]
    return [torch.stack([o[i] for o in objs]) for i in range(19)]
