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
        # per-group step counters of the flag-aware step (trunk + density head + B | colour branch | feature branch)
        # (two banks: a step reads one and writes the advanced counters to the other, objnerf_adamw_step_flags)
        self._banks = torch.zeros(2, 3, dtype=torch.int32, device=self.arena.params.device)
        self._bank = 0

    def step(self, grads: torch.Tensor, has_grad: Optional[torch.Tensor] = None, flags: Optional[torch.Tensor] = None):
        """flags: the iteration's early-return flag pair (device int32[2]).  With it, tensors whose loss terms were
        constants this iteration are skipped like parameters with .grad = None in torch.optim.AdamW (no decay, no moment
        update, their own step count) -- decided on the device, no host sync.  Without it every tensor of `has_grad`
        is stepped with one global step count."""
        # (arena.version -- the "parameters were written through raw pointers" counter autograd.py checks -- is advanced by
        # the ops-level writers themselves)
        if flags is not None:
            ops.adamw_step_flags(self.arena, grads, self.exp_avg, self.exp_avg_sq, has_grad, flags, self._banks, self._bank,
                                 self.lr, self.weight_decay, self.betas[0], self.betas[1], self.eps)
            self._bank ^= 1
            return
        self.step_count += 1
        ops.adamw_step(self.arena, grads, self.exp_avg, self.exp_avg_sq, has_grad, self.step_count, self.lr,
                       self.weight_decay, self.betas[0], self.betas[1], self.eps)

    @property
    def group_steps(self) -> torch.Tensor:
        """int32[3]: steps taken so far by (trunk + density head + B | colour branch | feature branch)."""
        return self._banks[self._bank]

    def zero_grad(self, set_to_none=True):
        pass        # gradients are written (not accumulated) by objnerf_train_step
