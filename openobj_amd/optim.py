"""AdamW over the parameter arena -- torch.optim.AdamW as train.py:78 configures it (lr, weight_decay;
betas (0.9, 0.999), eps 1e-8), one objnerf_adamw_step launch per step."""
from typing import Optional

import torch

from . import ops


class ArenaAdamW:
    def __init__(self, arena: ops.ParamArena, lr=1e-3, weight_decay=1e-2, betas=(0.9, 0.999), eps=1e-8):
        self.arena, self.lr, self.weight_decay, self.betas, self.eps = arena, lr, weight_decay, betas, eps
        self.reset_state()

    def reset_state(self):
        """utils.update_vmap adds a NEW param group whenever an object is added, i.e. fresh Adam
        moments (train.py:272-276, utils.py:59-61)."""
        self.exp_avg = torch.zeros_like(self.arena.params)
        self.exp_avg_sq = torch.zeros_like(self.arena.params)
        self.step_count = 0

    def step(self, grads: torch.Tensor, has_grad: Optional[torch.Tensor] = None):
        self.step_count += 1
        ops.adamw_step(self.arena, grads, self.exp_avg, self.exp_avg_sq, has_grad, self.step_count, self.lr,
                       self.weight_decay, self.betas[0], self.betas[1], self.eps)

    def zero_grad(self, set_to_none=True):
        pass        # gradients are written (not accumulated) by objnerf_train_step
